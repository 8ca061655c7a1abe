'use strict';
// JS shim over the N-API addon: keeps the reference's call shapes for the hot path.
//   pippenger(c, points, scalars)          - same signature / validation / messages as
//                                            src/abstract/curve.ts:863-905
//   multiplyUnsafeBatch(c, points, scalars) - array form of Point.multiplyUnsafe
//                                            (src/abstract/weierstrass.ts:915-928)
//   multiplyBaseBatch(c, scalars)          - array form of BASE.multiply (curve.ts:588-606)
//   ed25519VerifyBatch(items, zip215)      - array form of eddsa.verify (edwards.ts:942-989)
// `c` is any Point constructor with the reference's CurvePointCons surface (BASE, ZERO, Fp, Fn,
// fromAffine; curve.ts:159-195) that was registered with `register(c, curveId)`; points are
// instances of it (toAffine()).  Plain CommonJS + BigInt so it also loads on old Node.
const native = require('./noble_gpu.node');
const crypto = require('crypto');

const CURVE = { SECP256K1: 0, ED25519: 1, BLS12_381_G1: 2, BLS12_381_G2: 3 };
const registry = new Map();
let inited = false;
function init(device) { if (!inited) { native.init(device || 0); inited = true; } }
// one process, several GPUs: pippenger shards its points over `deviceIds` (ncg_msm_multi: one RCCL all-gather of
// window sums, combine on the first device); every other call runs on the first device
function initMulti(deviceIds) { if (!inited) { native.initMulti(deviceIds); inited = true; multiDevice = deviceIds.length > 1; } return deviceIds.length; }
function register(c, curveId) { registry.set(c, curveId); }

// BigInt <-> little-endian bytes through hex strings (utils.ts:498 numberToBytesLE / :456
// bytesToNumberLE use the same route): ~15x faster than shifting a BigInt byte by byte, and the
// marshalling is what an end-to-end call from JS spends its time in (SURVEY 8a gotcha 8).
function leBytes(n, len, out, off) {
  const be = Buffer.from(n.toString(16).padStart(2 * len, '0'), 'hex');
  for (let i = 0; i < len; i++) out[off + i] = be[len - 1 - i];
}
function leNumber(buf, off, len) {
  const be = Buffer.allocUnsafe(len);
  for (let i = 0; i < len; i++) be[i] = buf[off + len - 1 - i];
  return BigInt('0x' + be.toString('hex'));
}
function coordsOf(aff, isFp2) { return isFp2 ? [aff.x.c0, aff.x.c1, aff.y.c0, aff.y.c1] : [aff.x, aff.y]; }

// Field.isValid / isValidNot0 of the scalar field (modular.ts:925-936): a non-bigint is a TypeError, not `false`
function isValidScalar(Fn, s) {
  if (typeof s !== 'bigint') throw new TypeError('invalid field element: expected bigint, got ' + typeof s);
  return s >= 0n && s < Fn.ORDER;
}
function isValidScalarNot0(Fn, s) { return s !== 0n && isValidScalar(Fn, s); }
function validateMSMPoints(points, c) {         // curve.ts:390-395 (aarray: utils.ts:134-143)
  if (!Array.isArray(points)) throw new TypeError('"points" expected array, got type=' + typeof points);
  points.forEach((p, i) => { if (!(p instanceof c)) throw new Error('invalid point at index ' + i); });
}
function validateMSMScalars(scalars, field) {   // curve.ts:398-404
  if (!Array.isArray(scalars)) throw new Error('array of scalars expected');
  scalars.forEach((s, i) => { if (!isValidScalar(field, s)) throw new Error('invalid scalar at index ' + i); });
}
// Point.multiply / multiplyUnsafe argument errors: weierstrass.ts:904, :920; edwards.ts:561, :573
function scalarRangeError(id, not0) {
  if (id === CURVE.ED25519) return new RangeError('invalid scalar: expected ' + (not0 ? '1' : '0') + ' <= sc < curve.n');
  return new RangeError('invalid scalar: out of range');
}
// coordinates / scalars cross as packed little-endian bytes; the BigInt -> bytes step runs natively
// (napi_get_value_bigint_words), ~5x faster than the hex-string route it replaces
// Affine coordinates of every point with ONE field inversion (Montgomery's trick over Z, what the reference's own normalizeZ does,
// curve.ts:311-326): a point with Z != 1 - any result of add / double / multiplyUnsafe - costs p.toAffine() a BigInt inversion
// (~400 field products), 4 096 of them seconds.  Needs the reference's surface (p.Z, toAffine(invertedZ), Fp.mul / inv / eql / is0 /
// ONE: weierstrass.ts:768-790, edwards.ts:603-607); any other registered class, and arrays whose points all have Z = 1, take
// the plain path.
function affineAll(c, points) {
  const Fp = c.Fp, n = points.length;
  const plain = () => { const out = new Array(n); for (let i = 0; i < n; i++) out[i] = points[i].toAffine(); return out; };
  if (n < 2 || !Fp || typeof Fp.mul !== 'function' || typeof Fp.inv !== 'function' || typeof Fp.eql !== 'function' ||
      typeof Fp.is0 !== 'function' || Fp.ONE === undefined || points[0].Z === undefined) return plain();
  const pre = new Array(n);
  let acc = Fp.ONE, any = false;
  for (let i = 0; i < n; i++) {
    const z = points[i].Z;
    if (z === undefined || Fp.is0(z)) { pre[i] = undefined; continue; }   // ZERO: toAffine() answers by itself
    if (!any && !Fp.eql(z, Fp.ONE)) any = true;
    pre[i] = acc;
    acc = Fp.mul(acc, z);
  }
  if (!any) return plain();
  let inv = Fp.inv(acc);
  const out = new Array(n);
  for (let i = n - 1; i >= 0; i--) {
    if (pre[i] === undefined) { out[i] = points[i].toAffine(); continue; }
    out[i] = points[i].toAffine(Fp.mul(inv, pre[i]));
    inv = Fp.mul(inv, points[i].Z);
  }
  return out;
}
function marshalPoints(c, id, points) {
  const pb = native.pointBytes(id), isFp2 = id === CURVE.BLS12_381_G2, fb = pb / (isFp2 ? 4 : 2);
  const flat = new Array(points.length * (isFp2 ? 4 : 2));
  const aff = affineAll(c, points);
  let k = 0;
  for (let i = 0; i < aff.length; i++) {
    const a = aff[i];
    if (isFp2) { flat[k++] = a.x.c0; flat[k++] = a.x.c1; flat[k++] = a.y.c0; flat[k++] = a.y.c1; }
    else if (id === CURVE.ED25519) { flat[k++] = a.x % c.Fp.ORDER; flat[k++] = a.y % c.Fp.ORDER; }
    else { flat[k++] = a.x; flat[k++] = a.y; }
  }
  return native.packBigInts(flat, fb);
}
function marshalScalars(scalars) { return native.packBigInts(scalars, 32); }
function unmarshalPoint(c, id, buf, off, inf) {
  if (inf) return c.ZERO;
  const pb = native.pointBytes(id), isFp2 = id === CURVE.BLS12_381_G2, fb = pb / (isFp2 ? 4 : 2);
  const v = [];
  for (let j = 0; j < pb / fb; j++) v.push(leNumber(buf, off + j * fb, fb));
  return isFp2 ? c.fromAffine({ x: { c0: v[0], c1: v[1] }, y: { c0: v[2], c1: v[3] } }) : c.fromAffine({ x: v[0], y: v[1] });
}
// n points at the start of `out` (pb bytes each), their infinity flags at out[infOff + i]; `keep(i)` (optional) filters
function unmarshalPoints(c, id, out, n, infOff, keep) {
  const pb = native.pointBytes(id), isFp2 = id === CURVE.BLS12_381_G2, per = isFp2 ? 4 : 2, fb = pb / per;
  const v = native.unpackBigInts(out.subarray(0, n * pb), fb);     // all coordinates in one native call
  const res = new Array(n);
  for (let i = 0; i < n; i++) {
    if (keep && !keep(i)) { res[i] = null; continue; }
    if (out[infOff + i] === 1) { res[i] = c.ZERO; continue; }
    const o = i * per;
    res[i] = isFp2 ? c.fromAffine({ x: { c0: v[o], c1: v[o + 1] }, y: { c0: v[o + 2], c1: v[o + 3] } }) : c.fromAffine({ x: v[o], y: v[o + 1] });
  }
  return res;
}
function curveId(c) {
  const id = registry.get(c);
  if (id === undefined) throw new Error('noble-gpu: Point class not registered');
  return id;
}

// Packed columns (SURVEY 8a gotcha 8; VERDICT r02 #9): every BigInt that crosses N-API costs ~100 ns
// (napi_get_element + napi_get_value_bigint_words), i.e. 2^16 points + scalars = 196 608 values = most of an
// end-to-end pippenger.  Callers that keep their data in columns skip that entirely:
//   packPoints(c, points)  -> Uint8Array, affine x || y little-endian per point (the wire format of include/ncg.h;
//                             Fp2 as c0 || c1) - marshal ONCE, reuse across calls (or uploadPoints for a resident set)
//   packScalars(scalars)   -> Uint8Array, 32 bytes little-endian each; a BigUint64Array of 4 words per scalar is the
//                             same bytes on a little-endian host and is accepted as is
// pippenger / multiplyUnsafeBatch accept these in place of Point[] / BigInt[]; packed scalars are range-checked on the
// bytes (same 'invalid scalar at index i'), packed points are trusted like fromAffine trusts its caller.
function packPoints(c, points) { const id = curveId(c); validateMSMPoints(points, c); return marshalPoints(c, id, points); }
function packScalars(scalars, Fn) { if (Fn) validateMSMScalars(scalars, Fn); return marshalScalars(scalars); }
function asBytes(x) {
  if (x instanceof Uint8Array) return x;
  if (typeof BigUint64Array !== 'undefined' && x instanceof BigUint64Array) return new Uint8Array(x.buffer, x.byteOffset, x.byteLength);
  return null;
}
// ---- the same Point[] seen again: keep its device copy (VERDICT r04 #7; the reference's own advice for a fixed point set,
// curve.ts:907-918).  `pippenger(c, points, scalars)` with Point OBJECTS spends its time in points.map(toAffine) + BigInt
// packing (2^20 points: ~200 ms against 3.5 ms on the device).  The reference's instances are frozen (weierstrass.ts:703,
// edwards.ts:391) and its own precompute cache is keyed by object identity (curve.ts pointPrecomputes WeakMap), so the same
// rule is used here: an array whose elements are all frozen is uploaded ONCE as a resident set, remembered in a WeakMap keyed by
// the array together with a snapshot of its element references; a later call with the same array is a hit only if every
// element is still the identical object (one tight loop of === over the array, ~1 ns per point) - replacing, adding or removing
// an element is a miss and rebuilds the entry.  At most `maxSets` sets stay on the device (least recently used is freed).
const POINT_CACHE = { enabled: true, minPoints: 1024, maxSets: 4, requireDeepFrozen: false, hits: 0, misses: 0 };
const pointCache = new WeakMap();
let cacheLru = [];
let multiDevice = false;
function setPointCache(opts) { Object.assign(POINT_CACHE, opts || {}); if (!POINT_CACHE.enabled) clearPointCache(); return POINT_CACHE; }
function clearPointCache() { cacheLru.forEach((e) => e.set.free()); cacheLru = []; }
// a HIT only: the resident set of this very array (every element still the identical object), else null.  Nothing is validated,
// marshalled or uploaded here.
function cacheLookup(c, points) {
  if (!POINT_CACHE.enabled || multiDevice || points.length < POINT_CACHE.minPoints) return null;
  const e = pointCache.get(points);
  if (e === undefined) return null;
  if (e.set.handle !== null && e.c === c && e.snap.length === points.length) {
    const snap = e.snap, n = snap.length;
    let i = 0;
    while (i < n && snap[i] === points[i]) i++;
    if (i === n) {
      POINT_CACHE.hits++;
      if (cacheLru[cacheLru.length - 1] !== e) { cacheLru = cacheLru.filter((x) => x !== e); cacheLru.push(e); }
      return e.set;
    }
  }
  e.set.free(); cacheLru = cacheLru.filter((x) => x !== e); pointCache.delete(points);   // stale: the array changed
  return null;
}
// a MISS: upload `points` (already validated by the caller, AFTER the scalars and lengths were - ADVICE r05: a call that is about
// to throw must not pay an upload or leave a set resident) and remember the set.  Retention: the LRU list holds the entries
// strongly (the WeakMap alone would let the device memory outlive nothing, but the list keeps up to `maxSets` sets and a snapshot
// of their point references alive until they are evicted, `clearPointCache()` is called, or the array is seen changed).
// Only arrays of FROZEN instances are cached (the reference's classes: weierstrass.ts:703, edwards.ts:391).  The check is shallow: the
// Fp2 coordinates {c0, c1} of a G2 point are plain objects the reference never mutates (its field ops return new objects) but
// does not freeze either; `setPointCache({ requireDeepFrozen: true })` refuses such points unless their coordinates are frozen too.
function cacheInsert(c, id, points) {
  if (!POINT_CACHE.enabled || multiDevice || points.length < POINT_CACHE.minPoints) return null;
  const deep = POINT_CACHE.requireDeepFrozen;
  for (let i = 0; i < points.length; i++) {
    const p = points[i];
    if (!Object.isFrozen(p)) return null;   // mutable stand-in classes: never cached
    if (deep && p.X !== undefined && typeof p.X === 'object' && !(Object.isFrozen(p.X) && Object.isFrozen(p.Y) && Object.isFrozen(p.Z))) return null;
  }
  POINT_CACHE.misses++;
  init();
  const set = new PointSet(c, id, native.uploadPoints(id, marshalPoints(c, id, points), false, false), points.length);
  const e = { c, snap: points.slice(), set };
  pointCache.set(points, e);
  cacheLru.push(e);
  while (cacheLru.length > POINT_CACHE.maxSets) cacheLru.shift().set.free();
  return set;
}
function msmCall(c, id, points, pBytes, scalars, sBytes, set) {
  init();
  const out = set !== null ? native.msmResident(set.handle, sBytes === null ? scalars : sBytes)
    : native.msm(id, pBytes === null ? marshalPoints(c, id, points) : pBytes, sBytes === null ? marshalScalars(scalars) : sBytes);
  return unmarshalPoint(c, id, out, 0, out[out.length - 1] === 1);
}
function pippenger(c, points, scalars) {
  const id = curveId(c);
  const pb = native.pointBytes(id);
  const pBytes = asBytes(points), sBytes = asBytes(scalars);
  // argument checks in the reference's order (curve.ts:871-875): points, scalars, lengths - all of them before anything is uploaded
  let set = pBytes === null && Array.isArray(points) ? cacheLookup(c, points) : null;    // a hit is an array validated before
  if (pBytes === null) { if (set === null) validateMSMPoints(points, c); }
  else if (pBytes.length % pb) throw new Error('noble-gpu: packed points: expected a multiple of ' + pb + ' bytes');
  if (sBytes === null) validateMSMScalars(scalars, c.Fn);
  else if (sBytes.length % 32) throw new Error('array of scalars expected');
  const np = pBytes === null ? points.length : pBytes.length / pb, ns = sBytes === null ? scalars.length : sBytes.length / 32;
  if (np !== ns) throw new Error('arrays of points and scalars must have equal length');
  if (sBytes !== null) checkPackedScalars(sBytes, c.Fn);
  if (np === 0) return c.ZERO;      // curve.ts:878
  if (set === null && pBytes === null) set = cacheInsert(c, id, points);
  return msmCall(c, id, points, pBytes, scalars, sBytes, set);
}
// ---- the REDIRECT (VERDICT r05 #2; SURVEY 8b "switches to GPU above a size threshold"): with the 12-line patch of INTEGRATION.md in
// src/abstract/curve.ts, the reference's own `pippenger` export hands every call with at least `minPoints` points to the backend
// registered for the Point class - AFTER its own argument checks and its empty-input return - and runs its own loop below that.
//     gpu.install(await import('@noble/curves/abstract/curve.js'), [secp256k1.Point, bls12_381.G1.Point, ...], { minPoints })
// Every caller of the reference's export (its tests, its benchmarks, bls.ts, user code) then runs on the GPU unchanged.
// DEFAULT_MIN_POINTS: the measured crossover of the reference's loop against one redirected call from Point objects
// (profiles/r06_js_threshold.json; EPYC 9575F + MI355X, marshalling included) lies BELOW one point: the reference's loop runs
// ~(Fn.BITS / w) (n + 2^w) BigInt point additions - 4.8 ms (secp256k1) / 2.5 (ed25519) / 5.1 (G1) / 19.3 ms (G2) for ONE point,
// 87 / 52 / 108 / 384 ms for 256 - against 0.20 / 0.21 / 0.25 / 0.37 ms and 0.41 / 0.50 / 0.62 / 1.02 ms redirected.  So the
// default redirects every non-empty call; `minPoints` stays the caller's knob (a process that must not touch the device for
// small inputs, a GPU shared with other work).
const DEFAULT_MIN_POINTS = 1;
const STATS = { msmRedirected: 0, msmPointsRedirected: 0 };
function redirectedMsm(c, points, scalars) {      // arguments already validated by the reference (curve.ts:871-878)
  const id = curveId(c);
  STATS.msmRedirected++;
  STATS.msmPointsRedirected += points.length;
  let set = cacheLookup(c, points);
  if (set === null) set = cacheInsert(c, id, points);
  return msmCall(c, id, points, null, scalars, null, set);
}
function install(curveModule, classes, opts) {
  if (!curveModule || typeof curveModule.setMSMBackend !== 'function')
    throw new Error('noble-gpu: install: this abstract/curve.js has no setMSMBackend (apply the patch of INTEGRATION.md)');
  const minPoints = opts && opts.minPoints !== undefined ? opts.minPoints : DEFAULT_MIN_POINTS;
  if (!Number.isSafeInteger(minPoints) || minPoints < 1) throw new Error('noble-gpu: install: minPoints must be a positive integer');
  for (const c of classes) { curveId(c); curveModule.setMSMBackend(c, { minPoints, msm: redirectedMsm }); }
  return minPoints;
}
function uninstall(curveModule, classes) { for (const c of classes) curveModule.setMSMBackend(c, undefined); }
// ---- resident point sets (interleavedMSMUnsafe's pattern, curve.ts:907-959): upload once, then only the
// scalars cross per call.  `scalars` may be a BigInt[] (validated like the reference, read natively as
// 64-bit words) or a Uint8Array of packed 32-byte little-endian values (no marshalling at all).
class PointSet {
  constructor(c, id, handle, length) { this.c = c; this.id = id; this.handle = handle; this.length = length; }
  free() { if (this.handle !== null) { native.freePoints(this.handle); this.handle = null; } }
  // bls12-381: every point is known to be torsion-free (decoded by fromBytes, or verified at upload): pippengerResident
  // then splits the scalars along the curve endomorphism - same result, half (G1) / a quarter (G2) of the windows
  get inSubgroup() { return this.handle !== null && native.inSubgroup(this.handle); }
}
// opts.checkSubgroup (bls12-381): run p.isTorsionFree() (bls12-381.ts:567-577, :599-601) on every point once, on the
// device; pippenger accepts points outside the subgroup, so a set holding one simply keeps the generic path
function uploadPoints(c, points, opts) {
  const id = curveId(c);
  validateMSMPoints(points, c);
  init();
  const set = new PointSet(c, id, native.uploadPoints(id, marshalPoints(c, id, points), false, false), points.length);
  if (opts && opts.checkSubgroup && (id === CURVE.BLS12_381_G1 || id === CURVE.BLS12_381_G2) && points.length) native.verifySubgroup(set.handle);
  return set;
}
function uploadEncoded(c, bytes, zip215) {      // concatenated compressed encodings; decoded + validated on the device
  const id = curveId(c);
  init();
  const eb = { 0: 33, 1: 32, 2: 48, 3: 96 }[id];
  if (!(bytes instanceof Uint8Array) || bytes.length % eb) throw new Error('noble-gpu: expected a Uint8Array of ' + eb + '-byte encodings');
  return new PointSet(c, id, native.uploadPoints(id, bytes, true, !!zip215), bytes.length / eb);
}
// packed scalars (32 bytes little-endian each) must satisfy the same rule as BigInt ones: 0 <= s < Fn.ORDER
// (validateMSMScalars, curve.ts:398-404 / 'invalid scalar: out of range', weierstrass.ts:920) - checked here, on
// the bytes, so that no entry point can reach the device with an unreduced scalar
function checkPackedScalars(bytes, Fn) {
  const order = Fn.ORDER;
  const ob = new Uint8Array(32);
  for (let i = 0, v = order; i < 32; i++, v >>= 8n) ob[i] = Number(v & 0xffn);
  const n = bytes.length >>> 5;
  // fast path: the top 32-bit word decides for all but ~2^-32 of the scalars of a full-width order
  const top = ((ob[31] << 24) | (ob[30] << 16) | (ob[29] << 8) | ob[28]) >>> 0;
  const u32 = (bytes.byteOffset & 3) === 0 ? new Uint32Array(bytes.buffer, bytes.byteOffset, bytes.length >>> 2) : null;
  for (let i = 0; i < n; i++) {
    if (u32 !== null && u32[8 * i + 7] < top) continue;
    const o = i << 5;
    let k = 31;
    while (k >= 0 && bytes[o + k] === ob[k]) k--;
    if (k < 0 || bytes[o + k] > ob[k]) throw new Error('invalid scalar at index ' + i);
  }
}
function residentScalars(set, scalars) {
  if (scalars instanceof Uint8Array) {
    if (scalars.length !== 32 * set.length) throw new Error('arrays of points and scalars must have equal length');
    checkPackedScalars(scalars, set.c.Fn);
    return scalars;
  }
  validateMSMScalars(scalars, set.c.Fn);
  if (scalars.length !== set.length) throw new Error('arrays of points and scalars must have equal length');
  return scalars;
}
function pippengerResident(set, scalars) {
  const sc = residentScalars(set, scalars);
  if (set.length === 0) return set.c.ZERO;
  const out = native.msmResident(set.handle, sc);
  return unmarshalPoint(set.c, set.id, out, 0, out[out.length - 1] === 1);
}
function multiplyUnsafeBatchResident(set, scalars) {
  const sc = residentScalars(set, scalars);
  if (set.length === 0) return [];
  const n = set.length, pb = native.pointBytes(set.id);
  const out = native.mulVarResident(set.handle, sc);
  return unmarshalPoints(set.c, set.id, out, n, n * pb);
}
// interleavedMSMUnsafe (curve.ts:938-959): MSM over a FIXED point set, returns the closure scalars -> Point.
// Same argument checks and messages; the closure accepts at most points.length scalars, omitted trailing ones
// are zero.  windowSize only sizes the reference's wNAF tables and never changes the result: here the set
// is uploaded once and each call runs the bucket MSM on the resident points.
function interleavedMSMUnsafe(c, points, windowSize) {
  const bits = c.Fn.BITS;
  if (!Number.isSafeInteger(windowSize) || windowSize < 2 || windowSize > bits)
    throw new Error('invalid window size, expected [2..' + bits + '], got W=' + windowSize);
  validateMSMPoints(points, c);
  const n = points.length;
  const set = n ? uploadPoints(c, points) : null;
  if (set) native.precomputePoints(set.handle);   // the reference builds its per-point tables here (curve.ts:948)
  return (scalars) => {
    validateMSMScalars(scalars, c.Fn);
    if (scalars.length > n) throw new Error('array of scalars must not be larger than array of points');
    if (n === 0) return c.ZERO;
    const padded = scalars.length === n ? scalars : scalars.concat(new Array(n - scalars.length).fill(0n));
    const out = native.msmResident(set.handle, padded);
    return unmarshalPoint(c, set.id, out, 0, out[out.length - 1] === 1);
  };
}
// eddsa.verify for a batch from (sig, msg, publicKey): the SHA-512 challenge is computed on the device
function ed25519VerifyBatchDevice(items, zip215) {
  const n = items.length;
  if (n === 0) return [];
  let total = 0;
  items.forEach((it) => { total += it.msg.length; });
  const sig = new Uint8Array(64 * n), pk = new Uint8Array(32 * n), msgs = new Uint8Array(total), offs = new Uint8Array(8 * (n + 1));
  const dv = new DataView(offs.buffer);
  let pos = 0;
  items.forEach((it, i) => {
    if (it.sig.length !== 64) throw new Error('"signature" expected Uint8Array of length 64');
    if (it.publicKey.length !== 32) throw new Error('"publicKey" expected Uint8Array of length 32');
    sig.set(it.sig, 64 * i); pk.set(it.publicKey, 32 * i); msgs.set(it.msg, pos);
    dv.setUint32(8 * i, pos % 4294967296, true); dv.setUint32(8 * i + 4, Math.floor(pos / 4294967296), true);
    pos += it.msg.length;
  });
  dv.setUint32(8 * n, pos % 4294967296, true); dv.setUint32(8 * n + 4, Math.floor(pos / 4294967296), true);
  init();
  return Array.from(native.ed25519VerifyMsgs(sig, pk, msgs, offs, zip215 !== false)).map((x) => x === 1);
}

// secp256k1.verify(sig, msgHash, publicKey, { prehash: false, lowS }) for a batch (weierstrass.ts:1571-1620): compact
// 64-byte signatures, 32-byte message hashes (hash with the reference's own sha256 first for prehash: true), 33-byte
// compressed keys; key decompression, s^-1 mod n, u1 G + u2 P and the comparison run on the device
function ecdsaVerifyBatch(items, lowS) {
  const n = items.length;
  if (n === 0) return [];
  const sig = new Uint8Array(64 * n), hs = new Uint8Array(32 * n), pk = new Uint8Array(33 * n);
  const live = new Array(n).fill(true);
  items.forEach((it, i) => {
    if (!(it.sig instanceof Uint8Array) || it.sig.length !== 64) throw new Error('"signature" expected Uint8Array of length 64');
    if (!(it.msgHash instanceof Uint8Array) || it.msgHash.length !== 32) throw new Error('"msgHash" expected Uint8Array of length 32');
    if (!(it.publicKey instanceof Uint8Array)) throw new Error('"publicKey" expected Uint8Array');
    if (it.publicKey.length !== 33) { live[i] = false; pk[33 * i] = 2; return; }   // other encodings: compress them first
    sig.set(it.sig, 64 * i); hs.set(it.msgHash, 32 * i); pk.set(it.publicKey, 33 * i);
  });
  init();
  return Array.from(native.ecdsaVerify(sig, hs, pk, lowS !== false)).map((x, i) => live[i] && x === 1);
}

function packMsgs(items, n) {
  let total = 0;
  items.forEach((it) => { total += it.msg.length; });
  const msgs = new Uint8Array(total), offs = new Uint8Array(8 * (n + 1));
  const dv = new DataView(offs.buffer);
  let pos = 0;
  items.forEach((it, i) => {
    if (!(it.msg instanceof Uint8Array)) throw new Error('"message" expected Uint8Array');
    msgs.set(it.msg, pos);
    dv.setUint32(8 * i, pos % 4294967296, true); dv.setUint32(8 * i + 4, Math.floor(pos / 4294967296), true);
    pos += it.msg.length;
  });
  dv.setUint32(8 * n, pos % 4294967296, true); dv.setUint32(8 * n + 4, Math.floor(pos / 4294967296), true);
  return [msgs, offs];
}
// secp256k1.verify(sig, msg, publicKey, { lowS }) with the default prehash for a batch: SHA-256 of every message, key
// handling (all keys 33-byte compressed or all 65-byte uncompressed) and the verification run on the device
function ecdsaVerifyBatchMsgs(items, lowS) {
  const n = items.length;
  if (n === 0) return [];
  const kb = items[0].publicKey.length;
  if (kb !== 33 && kb !== 65) throw new Error('noble-gpu: ecdsaVerifyBatchMsgs: 33- or 65-byte public keys');
  const sig = new Uint8Array(64 * n), pk = new Uint8Array(kb * n);
  items.forEach((it, i) => {
    if (!(it.sig instanceof Uint8Array) || it.sig.length !== 64) throw new Error('"signature" expected Uint8Array of length 64');
    if (!(it.publicKey instanceof Uint8Array) || it.publicKey.length !== kb) throw new Error('noble-gpu: ecdsaVerifyBatchMsgs: mixed public-key lengths');
    sig.set(it.sig, 64 * i); pk.set(it.publicKey, kb * i);
  });
  const [msgs, offs] = packMsgs(items, n);
  init();
  return Array.from(native.secpVerifyMsgs(0, sig, pk, msgs, offs, lowS !== false)).map((x) => x === 1);
}
// schnorr.verify(sig, msg, publicKey) (src/secp256k1.ts:228-258) for a batch, tagged challenge hash on the device
function schnorrVerifyBatch(items) {
  const n = items.length;
  if (n === 0) return [];
  const sig = new Uint8Array(64 * n), pk = new Uint8Array(32 * n);
  items.forEach((it, i) => {
    if (!(it.sig instanceof Uint8Array) || it.sig.length !== 64) throw new Error('"signature" expected Uint8Array of length 64');
    if (!(it.publicKey instanceof Uint8Array) || it.publicKey.length !== 32) throw new Error('"publicKey" expected Uint8Array of length 32');
    sig.set(it.sig, 64 * i); pk.set(it.publicKey, 32 * i);
  });
  const [msgs, offs] = packMsgs(items, n);
  init();
  return Array.from(native.secpVerifyMsgs(1, sig, pk, msgs, offs, true)).map((x) => x === 1);
}
// recoverPublicKey for a batch of { sig (65 bytes: recovery id || r || s), msgHash (32 bytes) }: compressed keys, null
// where the reference throws
function ecdsaRecoverBatch(items) {
  const n = items.length;
  if (n === 0) return [];
  const sig = new Uint8Array(65 * n), hs = new Uint8Array(32 * n);
  items.forEach((it, i) => {
    if (!(it.sig instanceof Uint8Array) || it.sig.length !== 65) throw new Error('"signature" expected Uint8Array of length 65');
    if (!(it.msgHash instanceof Uint8Array) || it.msgHash.length !== 32) throw new Error('"msgHash" expected Uint8Array of length 32');
    sig.set(it.sig, 65 * i); hs.set(it.msgHash, 32 * i);
  });
  init();
  const out = native.ecdsaRecover(sig, hs);
  return items.map((_, i) => (out[33 * n + i] === 1 ? out.slice(33 * i, 33 * i + 33) : null));
}

function multiplyUnsafeBatch(c, points, scalars) {
  const id = curveId(c);
  validateMSMPoints(points, c);
  if (points.length !== scalars.length) throw new Error('arrays of points and scalars must have equal length');
  scalars.forEach((s) => { if (!isValidScalar(c.Fn, s)) throw scalarRangeError(id, false); });   // weierstrass.ts:920, edwards.ts:573
  if (points.length === 0) return [];
  init();
  const n = points.length, pb = native.pointBytes(id);
  const out = native.mulVarBatch(id, marshalPoints(c, id, points), marshalScalars(scalars));
  return unmarshalPoints(c, id, out, n, n * pb);
}
function multiplyBaseBatch(c, scalars) {
  const id = curveId(c);
  scalars.forEach((s) => { if (!isValidScalarNot0(c.Fn, s)) throw scalarRangeError(id, true); });   // weierstrass.ts:904, edwards.ts:561
  if (id === CURVE.ED25519) return multiplyUnsafeBatch(c, scalars.map(() => c.BASE), scalars);
  if (scalars.length === 0) return [];
  init();
  const n = scalars.length, pb = native.pointBytes(id);
  const out = native.mulBaseBatch(id, marshalScalars(scalars));
  return unmarshalPoints(c, id, out, n, n * pb);
}
const ED_L = 0x1000000000000000000000000000000014def9dea2f79cd65812631a5cf5d3edn;
function ed25519VerifyBatch(items, zip215) {   // items: [{sig, msg, publicKey}] of Uint8Array
  const n = items.length;
  const sig = new Uint8Array(64 * n), pk = new Uint8Array(32 * n), ks = new Uint8Array(32 * n);
  items.forEach((it, i) => {
    if (it.sig.length !== 64) throw new Error('"signature" expected Uint8Array of length 64');
    if (it.publicKey.length !== 32) throw new Error('"publicKey" expected Uint8Array of length 32');
    sig.set(it.sig, 64 * i); pk.set(it.publicKey, 32 * i);
    const h = crypto.createHash('sha512');     // edwards.ts:900-906 (SHA-512 is @noble/hashes in the reference)
    h.update(it.sig.subarray(0, 32)); h.update(it.publicKey); h.update(it.msg);
    leBytes(leNumber(h.digest(), 0, 64) % ED_L, 32, ks, 32 * i);
  });
  if (n === 0) return [];
  init();
  return Array.from(native.ed25519VerifyBatch(sig, pk, ks, zip215 !== false)).map((x) => x === 1);
}


// ---- codecs: array forms of Point.fromBytes / toBytes (compressed) --------------------------------
//   weierstrass.ts:541-605, bls12-381.ts:377-459 (+ subgroup checks :567-577, :599-601), edwards.ts:405-436,620-628
// fromBytesBatch returns null where the reference would throw.
const ENC = [33, 32, 48, 96];
function fromBytesBatch(c, encodings, zip215) {
  const id = curveId(c), eb = ENC[id], pb = native.pointBytes(id), n = encodings.length;
  const buf = new Uint8Array(n * eb);
  encodings.forEach((e, i) => {
    if (!(e instanceof Uint8Array) || e.length !== eb) throw new Error('invalid point encoding at index ' + i + ': expected ' + eb + ' bytes');
    buf.set(e, i * eb);
  });
  if (n === 0) return [];
  init();
  const out = native.decodePoints(id, buf, !!zip215);
  return unmarshalPoints(c, id, out, n, n * pb + n, (i) => out[n * pb + i] === 1);
}
function toBytesBatch(c, points) {
  const id = curveId(c), eb = ENC[id];
  validateMSMPoints(points, c);
  if (points.length === 0) return [];
  init();
  const out = native.encodePoints(id, marshalPoints(c, id, points));
  const n = points.length;
  return points.map((_, i) => {
    if (!out[n * eb + i]) throw new Error('bad point: ZERO');     // weierstrass.ts:545
    return out.slice(i * eb, (i + 1) * eb);
  });
}

// sum of the decoded points: the group part of bls.aggregatePublicKeys on encoded keys (abstract/bls.ts:857-873)
function aggregateFromBytes(c, encodings, zip215) {
  const id = curveId(c), eb = ENC[id], pb = native.pointBytes(id), n = encodings.length;
  if (!Array.isArray(encodings) || n === 0) throw new Error('expected non-empty array');   // bls.ts:426-431 aNonEmpty
  const buf = new Uint8Array(n * eb);
  encodings.forEach((e, i) => {
    if (!(e instanceof Uint8Array) || e.length !== eb) throw new Error('invalid point encoding at index ' + i + ': expected ' + eb + ' bytes');
    buf.set(e, i * eb);
  });
  init();
  const out = native.aggregateEncoded(id, buf, !!zip215);
  return unmarshalPoint(c, id, out, 0, out[pb] === 1);
}

// ---- FFT over the bls12-381 scalar field: FFT(roots, Fr).direct / .inverse (fft.ts:518-577) --------
const FR = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001n;
function powMod(b, e, m) { let r = 1n; b %= m; while (e > 0n) { if (e & 1n) r = r * b % m; b = b * b % m; e >>= 1n; } return r; }
function fftFr(values, opts) {                  // opts: { inverse, brpInput, brpOutput, generator = 7n }
  opts = opts || {};
  const N = values.length;
  if (N === 0 || (N & (N - 1)) !== 0) throw new Error('FFT: Polynomial size should be power of two');
  const bits = 31 - Math.clz32(N);
  const omega = powMod(opts.generator || 7n, (FR - 1n) >> BigInt(bits), FR);   // rootsOfUnity.omega(bits), fft.ts:238-241
  const data = new Uint8Array(N * 32), om = new Uint8Array(32);
  values.forEach((v, i) => {
    if (typeof v !== 'bigint' || v < 0n || v >= FR) throw new Error('invalid field element: outside of range 0..ORDER');
    leBytes(v, 32, data, 32 * i);
  });
  leBytes(omega, 32, om, 0);
  init();
  const out = native.ntt(bits, om, data, (opts.inverse ? 1 : 0) | (opts.brpInput ? 2 : 0) | (opts.brpOutput ? 4 : 0));
  return values.map((_, i) => leNumber(out, 32 * i, 32));
}

// ---- hash-to-curve for bls12-381 G1 / G2: createHasher(...).hashToCurve (hash-to-curve.ts:441-548) -
const BLS_P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaabn;
function expandMessageXmd(msg, dst, len) {      // hash-to-curve.ts:189-228 with SHA-256
  const sha = (...parts) => { const h = crypto.createHash('sha256'); parts.forEach((p) => h.update(p)); return h.digest(); };
  if (dst.length > 255) dst = sha(Buffer.from('H2C-OVERSIZE-DST-'), dst);
  const ell = Math.ceil(len / 32);
  if (len > 65535 || ell > 255) throw new Error('expand_message_xmd: invalid lenInBytes');
  const dstPrime = Buffer.concat([dst, Buffer.from([dst.length])]);
  const b0 = sha(Buffer.alloc(64), msg, Buffer.from([len >> 8, len & 255, 0]), dstPrime);
  const b = [sha(b0, Buffer.from([1]), dstPrime)];
  for (let i = 1; i < ell; i++) b.push(sha(Buffer.from(b0.map((x, j) => x ^ b[i - 1][j])), Buffer.from([i + 1]), dstPrime));
  return Buffer.concat(b).slice(0, len);
}
function hashToCurveBatch(c, msgs, DST) {
  const id = curveId(c);
  if (id !== CURVE.BLS12_381_G1 && id !== CURVE.BLS12_381_G2) throw new Error('noble-gpu: hashToCurveBatch: bls12-381 G1 / G2 only');
  const m = id === CURVE.BLS12_381_G2 ? 2 : 1, L = 64, count = 2;
  const dst = Buffer.from(DST || (m === 2 ? 'BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_NUL_' : 'BLS_SIG_BLS12381G1_XMD:SHA-256_SSWU_RO_NUL_'));
  const u = new Uint8Array(msgs.length * count * m * 48);
  msgs.forEach((msg, i) => {
    const prb = expandMessageXmd(Buffer.from(msg), dst, count * m * L);       // hash_to_field :312-378
    for (let e = 0; e < count * m; e++) {
      let v = 0n;
      for (let j = 0; j < L; j++) v = (v << 8n) | BigInt(prb[e * L + j]);
      leBytes(v % BLS_P, 48, u, (i * count * m + e) * 48);
    }
  });
  if (msgs.length === 0) return [];
  init();
  const pb = native.pointBytes(id), n = msgs.length;
  const out = native.mapToCurve(id, count, u);
  return unmarshalPoints(c, id, out, n, n * pb);
}

module.exports = { CURVE, init, initMulti, register, install, uninstall, STATS, DEFAULT_MIN_POINTS, setPointCache, clearPointCache, packPoints, packScalars, pippenger, multiplyUnsafeBatch, multiplyBaseBatch, ed25519VerifyBatch,
                   PointSet, uploadPoints, uploadEncoded, interleavedMSMUnsafe, pippengerResident, multiplyUnsafeBatchResident, ed25519VerifyBatchDevice, ecdsaVerifyBatch, ecdsaVerifyBatchMsgs, schnorrVerifyBatch, ecdsaRecoverBatch,
                   fromBytesBatch, toBytesBatch, aggregateFromBytes, fftFr, hashToCurveBatch, native };
