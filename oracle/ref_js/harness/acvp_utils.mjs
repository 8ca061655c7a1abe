// test/vectors/acvp-vectors is a git submodule that is absent from this checkout of the reference; test/utils.ts imports its
// `jsonGZ` reader.  TEST INFRASTRUCTURE (oracle/): the reader alone (gunzip + JSON.parse); tests that need the vectors themselves
// fail on the missing files and are filtered out by the driver script.
import { readFileSync } from 'fs';
import { gunzipSync } from 'zlib';
export function jsonGZ(path) { return JSON.parse(gunzipSync(readFileSync(path)).toString('utf8')); }
