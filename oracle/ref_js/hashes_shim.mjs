// Stand-in for the `@noble/hashes` entry points the reference's curve code imports (utils.js, sha2.js, hmac.js), on top of
// node:crypto - TEST INFRASTRUCTURE for oracle/_ref/refjs.bundle (oracle/ref_js/downlevel.py copies it there).  Only what the hot
// path's files touch: byte helpers and argument checks with the upstream argument meaning, SHA-256/384/512 and HMAC as
// callable hash objects (`h(msg)`, `h.create().update().digest()`, `.outputLen`, `.blockLen`).
import crypto from 'crypto';

export function isBytes(a) {
  return a instanceof Uint8Array || (ArrayBuffer.isView(a) && a.constructor.name === 'Uint8Array');
}
export function anumber(n, title = '') {
  if (!Number.isSafeInteger(n) || n < 0) {
    const prefix = title && `"${title}" `;
    throw new Error(`${prefix}expected integer >= 0, got ${n}`);
  }
}
export function abytes(value, length, title = '') {
  const bytes = isBytes(value);
  const len = value == null ? undefined : value.length;
  const needsLen = length !== undefined;
  if (!bytes || (needsLen && len !== length)) {
    const prefix = title && `"${title}" `;
    const ofLen = needsLen ? ` of length ${length}` : '';
    const got = bytes ? `length=${len}` : `type=${typeof value}`;
    throw new Error(prefix + 'expected Uint8Array' + ofLen + ', got ' + got);
  }
  return value;
}
const hexes = Array.from({ length: 256 }, (_, i) => i.toString(16).padStart(2, '0'));
export function bytesToHex(bytes) {
  abytes(bytes);
  let hex = '';
  for (let i = 0; i < bytes.length; i++) hex += hexes[bytes[i]];
  return hex;
}
export function hexToBytes(hex) {
  if (typeof hex !== 'string') throw new Error('hex string expected, got ' + typeof hex);
  if (hex.length % 2) throw new Error('hex string expected, got unpadded hex of length ' + hex.length);
  if (!/^[0-9a-fA-F]*$/.test(hex)) throw new Error('hex string expected, got non-hex character');
  const out = new Uint8Array(hex.length / 2);
  for (let i = 0; i < out.length; i++) out[i] = parseInt(hex.substr(2 * i, 2), 16);
  return out;
}
export function concatBytes(...arrays) {
  let sum = 0;
  for (const a of arrays) {
    abytes(a);
    sum += a.length;
  }
  const res = new Uint8Array(sum);
  let pad = 0;
  for (const a of arrays) {
    res.set(a, pad);
    pad += a.length;
  }
  return res;
}
export function randomBytes(n = 32) {
  anumber(n, 'bytesLength');
  return new Uint8Array(crypto.randomBytes(n));
}
export function utf8ToBytes(str) {
  if (typeof str !== 'string') throw new Error('string expected');
  return new Uint8Array(Buffer.from(str, 'utf8'));
}
export function clean(...arrays) {
  for (const a of arrays) a.fill(0);
}
export function ahash(h) {
  if (typeof h !== 'function' || typeof h.create !== 'function') throw new Error('Hash must wrapped by utils.createHasher');
  anumber(h.outputLen);
  anumber(h.blockLen);
}
function makeHash(name, outputLen, blockLen) {
  const create = () => {
    const c = crypto.createHash(name);
    const o = {
      outputLen,
      blockLen,
      update(m) {
        c.update(abytes(m));
        return o;
      },
      digest() {
        return new Uint8Array(c.digest());
      },
      destroy() {},
    };
    return o;
  };
  const h = (msg) => create().update(msg).digest();
  h.create = create;
  h.outputLen = outputLen;
  h.blockLen = blockLen;
  return h;
}
export const sha256 = makeHash('sha256', 32, 64);
export const sha384 = makeHash('sha384', 48, 128);
export const sha512 = makeHash('sha512', 64, 128);
export const sha224 = makeHash('sha224', 28, 64);
// SHAKE (ed448.ts builds its hashers from shake256 while the module loads; test/point.helpers.ts imports every curve): node:crypto's XOF with
// a fixed output length, noble's `(msg, { dkLen })` / `.create({ dkLen })` call shapes
function makeXof(name, blockLen) {
  const create = (opts) => {
    const outputLen = opts && opts.dkLen !== undefined ? opts.dkLen : 32;
    const parts = [];
    const o = {
      outputLen,
      blockLen,
      update(m) {
        parts.push(Uint8Array.from(abytes(m)));
        return o;
      },
      digest() {
        const c = crypto.createHash(name, { outputLength: outputLen });
        for (const p of parts) c.update(p);
        return new Uint8Array(c.digest());
      },
      destroy() {},
    };
    return o;
  };
  const h = (msg, opts) => create(opts).update(msg).digest();
  h.create = create;
  h.outputLen = 32;
  h.blockLen = blockLen;
  return h;
}
export const shake256 = makeXof('shake256', 136);
export const shake128 = makeXof('shake128', 168);
// createHasher(cons) of @noble/hashes/utils.js: a hash function object from a constructor of hash instances
export function createHasher(hashCons, info = {}) {
  const hashC = (msg, opts) => hashCons(opts).update(msg).digest();
  const tmp = hashCons(undefined);
  hashC.outputLen = tmp.outputLen;
  hashC.blockLen = tmp.blockLen;
  hashC.create = (opts) => hashCons(opts);
  Object.assign(hashC, info);
  return Object.freeze(hashC);
}
// misc.ts (jubjub / babyjubjub, not on the hot path) names two BLAKE hashes at module load; nothing the hot-path tests run calls them
function absentHash(name, outputLen, blockLen) {
  const fail = () => {
    throw new Error(name + ' is not part of the shim');
  };
  const h = () => fail();
  h.create = fail;
  h.outputLen = outputLen;
  h.blockLen = blockLen;
  return h;
}
export const blake512 = absentHash('blake512', 64, 128);
export const blake2s = absentHash('blake2s', 32, 64);
const nodeName = (h) => (h === sha256 ? 'sha256' : h === sha512 ? 'sha512' : h === sha384 ? 'sha384' : h === sha224 ? 'sha224' : null);
export function hmac(hash, key, message) {
  return hmac.create(hash, key).update(message).digest();
}
hmac.create = (hash, key) => {
  ahash(hash);
  const name = nodeName(hash);
  if (!name) throw new Error('hmac shim: unknown hash');
  const c = crypto.createHmac(name, abytes(key));
  const o = {
    outputLen: hash.outputLen,
    blockLen: hash.blockLen,
    update(m) {
      c.update(abytes(m));
      return o;
    },
    digest() {
      return new Uint8Array(c.digest());
    },
    destroy() {},
  };
  return o;
};
