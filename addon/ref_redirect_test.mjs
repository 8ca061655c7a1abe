// The REDIRECT under the reference's own tests and benchmarks (VERDICT r05 #2).
//
// `js_hooked/` of oracle/_ref/refjs.bundle is the reference with the 12-line MSM-backend patch of INTEGRATION.md applied to
// src/abstract/curve.ts (oracle/ref_js/downlevel.py --gpu-hook) - nothing else differs from the reference's text.  This script
// registers the reference's Point classes with the shim, calls `gpu.install(curveModule, classes, { minPoints })`, and then runs,
// UNMODIFIED, with the GPU underneath the reference's own `pippenger` export:
//   A. benchmark/msm_timings.ts          (whole file: its own `check` compares every msm() with a sum of multiplyUnsafe)
//   B. benchmark/bls12-381.ts:64-79      (the file with a name filter on its bench() calls: 32 768 points, `MSM pippenger x32768`;
//                                         the result is compared with the reference's loop after uninstalling the backend)
//   C. test/point.test.ts                (the file; the tests of secp256k1, ed25519, bls12_381_G1, bls12_381_G2 + the toy-curve
//                                         and secp256k1 pippenger blocks :264-305, :685-706, :812-851 selected by NAME)
//   D. ed25519.verify over test/vectors/ed25519/vectors.txt (1024 lines) - batch verification on the device against the
//                                         reference's verify, line by line
//   E. the crossover table behind DEFAULT_MIN_POINTS (reference loop vs redirected call, n = 1 .. 256) -> JSON on stdout
// `gpu.STATS.msmRedirected` proves which calls took the GPU path; with minPoints above the input size the same tests must run
// the reference's own loop (counter unchanged).
//
//   node addon/ref_redirect_test.mjs <unpacked js_hooked dir> [--threshold-table out.json]
// TEST INFRASTRUCTURE (run by tests/test_node_redirect.py).
import assert from 'assert';
import fs from 'fs';
import path from 'path';
import { createRequire } from 'module';
import { pathToFileURL } from 'url';

const require = createRequire(import.meta.url);
const gpu = require('./noble_gpu.js');
const dir = path.resolve(process.argv[2] || '');
const tableOut = process.argv.includes('--threshold-table') ? process.argv[process.argv.indexOf('--threshold-table') + 1] : null;
const load = (f) => import(pathToFileURL(path.join(dir, f)).href);
const sleep = (ms) => new Promise((r) => setTimeout(r, ms));

async function main() {
  await load('polyfill.mjs');
  const curveMod = await load('src/abstract/curve.mjs');
  assert.strictEqual(typeof curveMod.setMSMBackend, 'function', 'the hooked abstract/curve.mjs exports setMSMBackend');
  const { pippenger } = curveMod;
  const { secp256k1 } = await load('src/secp256k1.mjs');
  const { bls12_381 } = await load('src/bls12-381.mjs');
  const { ed25519 } = await load('src/ed25519.mjs');
  const classes = [secp256k1.Point, ed25519.Point, bls12_381.G1.Point, bls12_381.G2.Point];
  const ids = [gpu.CURVE.SECP256K1, gpu.CURVE.ED25519, gpu.CURVE.BLS12_381_G1, gpu.CURVE.BLS12_381_G2];
  classes.forEach((c, i) => gpu.register(c, ids[i]));

  // install() refuses a module without the patch, and a class that was never registered
  assert.throws(() => gpu.install({}, classes), /no setMSMBackend/);
  assert.throws(() => gpu.install(curveMod, [class Foo {}]), /not registered/);
  assert.throws(() => gpu.install(curveMod, classes, { minPoints: 0 }), /positive integer/);

  let haveGpu = true;
  try { gpu.init(0); } catch (e) { haveGpu = false; console.log('no GPU here:', e.message); }

  // ---- without a GPU: the hook is inert until a backend is installed, and the reference's argument errors come first ----
  const G1 = bls12_381.G1.Point;
  const small = [G1.BASE, G1.BASE.double(), G1.BASE.negate()];
  const want0 = pippenger(G1, small, [5n, 7n, 11n]);
  gpu.install(curveMod, classes, { minPoints: 1000 });           // above the input: the reference's loop runs
  assert.ok(pippenger(G1, small, [5n, 7n, 11n]).equals(want0));
  assert.strictEqual(gpu.STATS.msmRedirected, 0, 'below minPoints nothing is redirected');
  gpu.install(curveMod, classes, { minPoints: 1 });
  assert.throws(() => pippenger(G1, [G1.BASE], [G1.Fn.ORDER]), /invalid scalar at index 0/);      // the reference's own checks, before the backend
  assert.throws(() => pippenger(G1, [G1.BASE, 5], [1n, 2n]), /invalid point at index 1/);
  assert.throws(() => pippenger(G1, [G1.BASE], [1n, 2n]), /arrays of points and scalars must have equal length/);
  assert.strictEqual(pippenger(G1, [], []), G1.ZERO);                                               // curve.ts:878, before the backend
  assert.strictEqual(gpu.STATS.msmRedirected, 0);
  if (!haveGpu) {
    assert.throws(() => pippenger(G1, small, [5n, 7n, 11n]), /noble-gpu|GPU|device|hip/i);           // no CPU fallback behind the backend
    gpu.uninstall(curveMod, classes);
    assert.ok(pippenger(G1, small, [5n, 7n, 11n]).equals(want0), 'uninstall restores the reference loop');
    console.log('redirect: hook checks OK (no GPU)');
    return;
  }
  assert.ok(pippenger(G1, small, [5n, 7n, 11n]).equals(want0));
  assert.strictEqual(gpu.STATS.msmRedirected, 1, 'a 3-point call with minPoints = 1 took the GPU path');

  // ---- A. benchmark/msm_timings.ts, unmodified: 12 msm() cases checked against sums of multiplyUnsafe, then timed ----
  const benchMod = await load('harness/jsbt_bench.mjs');
  let before = gpu.STATS.msmRedirected;
  let rejected = null;
  const onRej = (e) => { rejected = e; };
  process.on('unhandledRejection', onRej);
  await load('benchmark/msm_timings.mjs');                         // an async IIFE: finished when its last section has its 10 rows
  for (let i = 0; i < 3000 && !rejected; i++) {
    if (benchMod.results.filter((r) => r.section === 'basic multiply').length === 10) break;
    await sleep(20);
  }
  if (rejected) throw rejected;
  const rowsA = benchMod.results.slice();
  assert.strictEqual(rowsA.filter((r) => r.section === 'single point').length, 5);
  assert.strictEqual(rowsA.filter((r) => r.section === 'multi point').length, 7);
  // check() ran 12 msm() once each (all through the backend: 1 and 5 points >= minPoints 1); the timing loops many more
  assert.ok(gpu.STATS.msmRedirected - before >= 12 + 12, 'msm_timings ran on the GPU: ' + (gpu.STATS.msmRedirected - before) + ' redirected calls');
  console.log('A msm_timings.ts: its own check passed with ' + (gpu.STATS.msmRedirected - before) + ' redirected msm() calls');

  // ---- B. benchmark/bls12-381.ts:64-79 (32 768 points, scalars 2^241 + ...; the `.map((i) => ...)` of the file makes them identical) ----
  before = gpu.STATS.msmRedirected;
  const pointsBefore = gpu.STATS.msmPointsRedirected;
  process.env.NCG_JSBT_FILTER = '^(initializing 32768 G1 points|MSM pippenger x32768)$';
  benchMod.results.length = 0;
  await load('benchmark/bls12-381.mjs');
  for (let i = 0; i < 30000 && !rejected; i++) {
    if (benchMod.results.some((r) => /^pairing/.test(r.name))) break;   // the file's last bench() call
    await sleep(20);
  }
  await sleep(200);
  if (rejected) throw rejected;
  delete process.env.NCG_JSBT_FILTER;
  const rowB = benchMod.results.find((r) => r.name === 'MSM pippenger x32768');
  assert.ok(rowB && rowB.runs >= 1);
  assert.ok(gpu.STATS.msmRedirected - before >= 1 && gpu.STATS.msmPointsRedirected - pointsBefore >= 32768);
  {  // the same call by hand: GPU (through the export) against the reference's loop (backend removed) on 32 768 identical points
    const amount = 32768, p1 = 2n ** 235n, p2 = 2n ** 241n;
    const P = G1.BASE.multiply(p1);
    const pts = Array(amount).fill(P), sc = Array(amount).fill(p2);
    const got = pippenger(G1, pts, sc);
    assert.ok(got.equals(P.multiply((p2 * BigInt(amount)) % G1.Fn.ORDER)), 'bls12-381.ts MSM value');
    gpu.uninstall(curveMod, classes);
    const t0 = Date.now();
    const ref = pippenger(G1, pts.slice(0, 4096), sc.slice(0, 4096));
    const refMs = Date.now() - t0;
    gpu.install(curveMod, classes, { minPoints: 1 });
    assert.ok(pippenger(G1, pts.slice(0, 4096), sc.slice(0, 4096)).equals(ref), '4096 of them: GPU == reference loop');
    console.log('B bls12-381.ts MSM pippenger x32768: ' + rowB.ms_per_op.toFixed(1) + ' ms/op redirected (' + rowB.runs + ' runs); the reference loop on 4 096 of them: ' + refMs + ' ms');
  }

  // ---- C. test/point.test.ts: the hot-path curves' tests + the pippenger blocks, selected by name ----
  before = gpu.STATS.msmRedirected;
  const { runCollected } = await load('harness/jsbt_test.mjs');
  await load('test/point.test.mjs');
  const re = /^basic curve tests > (basic curve )?(secp256k1|ed25519|bls12_381_G1|bls12_381_G2)( >|$)|toy curve, exhaustive > pippenger|real curves > pippenger/;
  const t0 = Date.now();
  const resC = await runCollected({ filter: re });
  assert.strictEqual(resC.failed, 0, JSON.stringify(resC.failures).slice(0, 3000));
  assert.strictEqual(resC.passed, 36, 'selected tests of test/point.test.ts: ' + resC.passed);
  const msmTests = resC.names.filter((n) => /multiscalar multiplication|pippenger/.test(n));
  assert.strictEqual(msmTests.length, 6);
  const redirectedC = gpu.STATS.msmRedirected - before;
  // each of the four 'basic, random, and precomputed MSM' tests calls pippenger >= 3 times with >= 1 point + 2 x 5 random runs
  assert.ok(redirectedC >= 4 * 8, 'point.test.ts MSM blocks ran on the GPU: ' + redirectedC);
  console.log('C test/point.test.ts: ' + resC.passed + ' tests passed (' + msmTests.length + ' MSM tests), ' + redirectedC + ' pippenger calls redirected, ' + (Date.now() - t0) + ' ms');
  // the same selection with the threshold above every input: the reference's loop, counter unchanged
  gpu.install(curveMod, classes, { minPoints: 1 << 20 });
  before = gpu.STATS.msmRedirected;
  const resC2 = await runCollected({ filter: /multiscalar multiplication|toy curve, exhaustive > pippenger/ });
  assert.strictEqual(resC2.failed, 0);
  assert.strictEqual(gpu.STATS.msmRedirected, before, 'above the threshold nothing is redirected');
  gpu.install(curveMod, classes, { minPoints: 1 });

  // ---- D. ed25519.verify over test/vectors/ed25519/vectors.txt ----
  {
    const hex = (s) => Uint8Array.from(Buffer.from(s, 'hex'));
    const lines = fs.readFileSync(path.join(dir, 'test/vectors/ed25519/vectors.txt'), 'utf8').trim().split('\n').map((l) => l.split(':'));
    assert.strictEqual(lines.length, 1024);
    const items = [], expect = [];
    lines.forEach((v, i) => {
      const pk = hex(v[1]), msg = hex(v[2]), sig = hex(v[3].slice(0, 128));
      items.push({ sig, msg, publicKey: pk });
      expect.push(true);
      if (i % 4 === 0) {            // a corrupted twin of every fourth line
        const bad = Uint8Array.from(sig);
        bad[i % 64] ^= 1 << (i % 8);
        items.push({ sig: bad, msg, publicKey: pk });
        expect.push(null);
      }
    });
    const t1 = Date.now();
    const got = gpu.ed25519VerifyBatchDevice(items, true);
    const gpuMs = Date.now() - t1;
    const t2 = Date.now();
    items.forEach((it, i) => {
      let ref;
      try { ref = ed25519.verify(it.sig, it.msg, it.publicKey); } catch (e) { ref = false; }
      if (expect[i] === true) assert.strictEqual(ref, true, 'vectors.txt line verifies in the reference');
      assert.strictEqual(got[i], ref, 'ed25519.verify #' + i);
    });
    console.log('D ed25519 vectors.txt: ' + items.length + ' verifications (1024 lines + 256 corrupted) equal to ed25519.verify; device batch ' + gpuMs + ' ms, reference ' + (Date.now() - t2) + ' ms');
  }

  // ---- E. crossover: the reference's loop against the redirected call on fresh Point objects (no cache: minPoints of the cache is 1024) ----
  if (tableOut) {
    const table = [];
    for (const [name, P] of [['secp256k1', secp256k1.Point], ['ed25519', ed25519.Point], ['bls12_381.G1', bls12_381.G1.Point], ['bls12_381.G2', bls12_381.G2.Point]]) {
      const N = P.Fn.ORDER;
      const base = [];
      for (let i = 0; i < 256; i++) base.push(P.BASE.multiply(BigInt(i) * 0x9e3779b97f4a7c15n % N + 1n));
      for (const n of [1, 2, 4, 8, 16, 32, 64, 256]) {
        const pts = base.slice(0, n), sc = pts.map((_, i) => (N - 1n - BigInt(i) * 0x1234567n) % N);
        const time = (fn, minMs) => { let k = 0; const t = process.hrtime.bigint(); let el = 0; do { fn(); k++; el = Number(process.hrtime.bigint() - t) / 1e6; } while (el < minMs && k < 200); return el / k; };
        gpu.uninstall(curveMod, classes);
        const want = pippenger(P, pts, sc);
        const refMs = time(() => pippenger(P, pts, sc), 60);
        gpu.install(curveMod, classes, { minPoints: 1 });
        assert.ok(pippenger(P, pts, sc).equals(want));
        const gpuMs = time(() => pippenger(P, pts, sc), 60);
        table.push({ curve: name, n, reference_loop_ms: +refMs.toFixed(3), redirected_ms: +gpuMs.toFixed(3), speedup: +(refMs / gpuMs).toFixed(1) });
      }
    }
    fs.writeFileSync(tableOut, JSON.stringify({ what: 'pippenger(c, points, scalars) through the reference\'s own export: its BigInt loop against the redirected call (Point objects in, Point out; marshalling included), one box', default_min_points: gpu.DEFAULT_MIN_POINTS, rows: table }, null, 1));
    console.log('E crossover table -> ' + tableOut);
    for (const r of table) if (r.n <= 8 || r.n === 256) console.log('  ' + r.curve + ' n=' + r.n + ': reference ' + r.reference_loop_ms + ' ms, redirected ' + r.redirected_ms + ' ms');
  }
  process.removeListener('unhandledRejection', onRej);
  gpu.uninstall(curveMod, classes);
  console.log('reference redirect OK: ' + gpu.STATS.msmRedirected + ' pippenger calls ran on the GPU (' + gpu.STATS.msmPointsRedirected + ' points)');
}
main().catch((e) => { console.error(e && e.stack ? e.stack : e); process.exit(1); });
