'use strict';
// End-to-end timing from the JS side (SURVEY 8d): BigInt marshalling + N-API + H2D/D2H + kernels, next to the
// native call alone, for pippenger and multiplyUnsafeBatch on secp256k1.  node addon/bench_js.js [log2n]
const gpu = require('./noble_gpu.js');
const N = 0xfffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364141n;
const P = 0xfffffffffffffffffffffffffffffffffffffffffffffffffffffffefffffc2fn;
class Point {
  constructor(x, y, inf) { this.x = x; this.y = y; this.inf = !!inf; Object.freeze(this); }   // frozen like the reference's instances (weierstrass.ts:703)
  static fromAffine(a) { return (a.x === 0n && a.y === 0n) ? Point.ZERO : new Point(a.x, a.y); }
  toAffine() { return this.inf ? { x: 0n, y: 0n } : { x: this.x, y: this.y }; }
}
Point.ZERO = new Point(0n, 0n, true);
Point.Fp = { ORDER: P, BYTES: 32 };
Point.Fn = { ORDER: N, BYTES: 32 };
gpu.register(Point, gpu.CURVE.SECP256K1);
gpu.init(0);
const n = 1 << (parseInt(process.argv[2] || '14', 10));
let s = 0x9e3779b97f4a7c15n;
const rnd = () => { s ^= s << 13n; s &= (1n << 64n) - 1n; s ^= s >> 7n; s ^= s << 17n; s &= (1n << 64n) - 1n; return s; };
const ks = [], ss = [];
for (let i = 0; i < n; i++) { ks.push(((rnd() << 64n) | rnd()) % (N - 1n) + 1n); ss.push(((rnd() << 192n) | (rnd() << 128n) | (rnd() << 64n) | rnd()) % N); }
const pts = gpu.multiplyBaseBatch(Point, ks);
const ms = (f) => { const t0 = process.hrtime.bigint(); const r = f(); return [Number(process.hrtime.bigint() - t0) / 1e6, r]; };
gpu.pippenger(Point, pts.slice(0, 64), ss.slice(0, 64));
// reference-shaped call, Point[] + BigInt[]: the FIRST call on an array marshals it (toAffine + packing) and leaves a device copy
// behind; later calls on the same array of the same frozen objects skip all of that (noble_gpu.js cachedSet)
gpu.setPointCache({ enabled: false });
const [tMsmNoCache] = ms(() => gpu.pippenger(Point, pts, ss));
gpu.setPointCache({ enabled: true });
const [tMsm] = ms(() => gpu.pippenger(Point, pts, ss));
const [tMul] = ms(() => gpu.multiplyUnsafeBatch(Point, pts, ss));
const best = (f, reps) => { let b = Infinity, r; for (let i = 0; i < reps; i++) { const [t, v] = ms(f); if (t < b) b = t; r = v; } return [b, r]; };
// native part alone: pre-marshalled buffers
const pb = new Uint8Array(n * 64), sb = new Uint8Array(n * 32);
const le = (v, len, out, off) => { for (let i = 0; i < len; i++) { out[off + i] = Number(v & 0xffn); v >>= 8n; } };
pts.forEach((p, i) => { le(p.x, 32, pb, 64 * i); le(p.y, 32, pb, 64 * i + 32); });
ss.forEach((v, i) => le(v, 32, sb, 32 * i));
gpu.native.msm(0, pb, sb);
const [tMsmN] = ms(() => gpu.native.msm(0, pb, sb));
const [tMulN] = ms(() => gpu.native.mulVarBatch(0, pb, sb));
// resident point set (uploadPoints once): only scalars cross - as BigInt[] (native 64-bit-word reads) and as packed bytes
// packed columns (packPoints once, scalars as BigUint64Array): the reference-shaped call without per-value N-API work
const packedPts = gpu.packPoints(Point, pts);
const sc64 = new BigUint64Array(sb.buffer);
const rFirst = gpu.pippenger(Point, pts, ss);
const [tCachedBig, rCb] = best(() => gpu.pippenger(Point, pts, ss), 3);          // cached Point[] + BigInt[] scalars
const [tCachedTyped, rCt] = best(() => gpu.pippenger(Point, pts, sc64), 5);      // cached Point[] + BigUint64Array scalars
const [tCachedBytes] = best(() => gpu.pippenger(Point, pts, sb), 5);             // ... + packed bytes
if (rCb.x !== rFirst.x || rCb.y !== rFirst.y || rCt.x !== rFirst.x || rCt.y !== rFirst.y) throw new Error('cached pippenger differs');
const cacheStats = gpu.setPointCache();
gpu.clearPointCache();
const [tPacked, rPacked] = ms(() => gpu.pippenger(Point, packedPts, sc64));
const rRef = gpu.pippenger(Point, pts, ss);
if (rPacked.x !== rRef.x || rPacked.y !== rRef.y) throw new Error('packed pippenger differs');
// the native calls on input buffers pinned once (native.hostRegister): no per-call page locking of the 96 MB at 2^20
gpu.native.hostRegister(pb);
gpu.native.hostRegister(sb);
if (packedPts instanceof Uint8Array && packedPts.buffer !== pb.buffer) gpu.native.hostRegister(packedPts);
const [tMsmPinned] = best(() => gpu.native.msm(0, pb, sb), 5);
const [tMulPinned] = best(() => gpu.native.mulVarBatch(0, pb, sb), 3);
// ... and with the OUTPUT array kept by the caller and pinned once too (mulVarBatch's optional 4th argument)
const outKeep = new Uint8Array(n * 65);
gpu.native.hostRegister(outKeep);
const [tMulPinnedOut, rKeep] = best(() => gpu.native.mulVarBatch(0, pb, sb, outKeep), 3);
const rFresh = gpu.native.mulVarBatch(0, pb, sb);
for (let i = 0; i < rFresh.length; i += 4099) if (rFresh[i] !== rKeep[i]) throw new Error('mulVarBatch(out) differs');
gpu.native.hostUnregister(outKeep);
const [tPackedPinned] = best(() => gpu.pippenger(Point, packedPts, sc64), 5);
gpu.native.hostUnregister(pb);
gpu.native.hostUnregister(sb);
if (packedPts instanceof Uint8Array && packedPts.buffer !== pb.buffer) gpu.native.hostUnregister(packedPts);
const set = gpu.uploadPoints(Point, pts);
gpu.pippengerResident(set, ss);
const [tResBig] = ms(() => gpu.pippengerResident(set, ss));
const [tResBytes] = ms(() => gpu.pippengerResident(set, sb));
set.free();
console.log(JSON.stringify({ n, pippenger_js_first_call_ms: tMsm, pippenger_js_no_cache_ms: tMsmNoCache, pippenger_js_cached_points_bigint_scalars_ms: tCachedBig, pippenger_js_cached_points_typed_scalars_ms: tCachedTyped, pippenger_js_cached_points_byte_scalars_ms: tCachedBytes, cache_hits: cacheStats.hits, cache_misses: cacheStats.misses, pippenger_native_pinned_ms: tMsmPinned, multiplyUnsafeBatch_native_pinned_ms: tMulPinned, multiplyUnsafeBatch_native_pinned_out_ms: tMulPinnedOut, pippenger_packed_columns_pinned_ms: tPackedPinned, pippenger_resident_bigint_ms: tResBig, pippenger_resident_bytes_ms: tResBytes, pippenger_js_ms: tMsm, pippenger_packed_columns_ms: tPacked, pippenger_native_ms: tMsmN, multiplyUnsafeBatch_js_ms: tMul,
  multiplyUnsafeBatch_native_ms: tMulN }));
