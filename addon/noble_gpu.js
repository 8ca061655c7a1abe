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
function register(c, curveId) { registry.set(c, curveId); }

function leBytes(n, len, out, off) {            // utils.ts:498 numberToBytesLE
  for (let i = 0; i < len; i++) { out[off + i] = Number(n & 0xffn); n >>= 8n; }
}
function leNumber(buf, off, len) {              // utils.ts:456 bytesToNumberLE
  let n = 0n;
  for (let i = len - 1; i >= 0; i--) n = (n << 8n) | BigInt(buf[off + i]);
  return n;
}
function coordsOf(aff, isFp2) { return isFp2 ? [aff.x.c0, aff.x.c1, aff.y.c0, aff.y.c1] : [aff.x, aff.y]; }

function validateMSMPoints(points, c) {         // curve.ts:390-395
  if (!Array.isArray(points)) throw new Error('array expected');
  points.forEach((p, i) => { if (!(p instanceof c)) throw new Error('invalid point at index ' + i); });
}
function validateMSMScalars(scalars, field) {   // curve.ts:398-404
  if (!Array.isArray(scalars)) throw new Error('array of scalars expected');
  scalars.forEach((s, i) => {
    if (typeof s !== 'bigint' || s < 0n || s >= field.ORDER) throw new Error('invalid scalar at index ' + i);
  });
}
function marshalPoints(c, id, points) {
  const pb = native.pointBytes(id), isFp2 = id === CURVE.BLS12_381_G2, fb = pb / (isFp2 ? 4 : 2);
  const buf = new Uint8Array(points.length * pb);
  points.forEach((p, i) => {
    coordsOf(p.toAffine(), isFp2).forEach((v, j) => leBytes(id === CURVE.ED25519 ? v % c.Fp.ORDER : v, fb, buf, i * pb + j * fb));
  });
  return buf;
}
function marshalScalars(scalars) {
  const buf = new Uint8Array(scalars.length * 32);
  scalars.forEach((s, i) => leBytes(s, 32, buf, 32 * i));
  return buf;
}
function unmarshalPoint(c, id, buf, off, inf) {
  if (inf) return c.ZERO;
  const pb = native.pointBytes(id), isFp2 = id === CURVE.BLS12_381_G2, fb = pb / (isFp2 ? 4 : 2);
  const v = [];
  for (let j = 0; j < pb / fb; j++) v.push(leNumber(buf, off + j * fb, fb));
  return isFp2 ? c.fromAffine({ x: { c0: v[0], c1: v[1] }, y: { c0: v[2], c1: v[3] } }) : c.fromAffine({ x: v[0], y: v[1] });
}
function curveId(c) {
  const id = registry.get(c);
  if (id === undefined) throw new Error('noble-gpu: Point class not registered');
  return id;
}

function pippenger(c, points, scalars) {
  const id = curveId(c);
  validateMSMPoints(points, c);
  validateMSMScalars(scalars, c.Fn);
  if (points.length !== scalars.length) throw new Error('arrays of points and scalars must have equal length');
  if (points.length === 0) return c.ZERO;      // curve.ts:878
  init();
  const out = native.msm(id, marshalPoints(c, id, points), marshalScalars(scalars));
  return unmarshalPoint(c, id, out, 0, out[out.length - 1] === 1);
}
function multiplyUnsafeBatch(c, points, scalars) {
  const id = curveId(c);
  validateMSMPoints(points, c);
  if (points.length !== scalars.length) throw new Error('arrays of points and scalars must have equal length');
  scalars.forEach((s) => {                     // weierstrass.ts:920
    if (typeof s !== 'bigint' || s < 0n || s >= c.Fn.ORDER) throw new RangeError('invalid scalar: out of range');
  });
  if (points.length === 0) return [];
  init();
  const n = points.length, pb = native.pointBytes(id);
  const out = native.mulVarBatch(id, marshalPoints(c, id, points), marshalScalars(scalars));
  return points.map((_, i) => unmarshalPoint(c, id, out, i * pb, out[n * pb + i] === 1));
}
function multiplyBaseBatch(c, scalars) {
  const id = curveId(c);
  scalars.forEach((s) => {                     // weierstrass.ts:904
    if (typeof s !== 'bigint' || s < 1n || s >= c.Fn.ORDER) throw new RangeError('invalid scalar: out of range');
  });
  if (id === CURVE.ED25519) return multiplyUnsafeBatch(c, scalars.map(() => c.BASE), scalars);
  if (scalars.length === 0) return [];
  init();
  const n = scalars.length, pb = native.pointBytes(id);
  const out = native.mulBaseBatch(id, marshalScalars(scalars));
  return scalars.map((_, i) => unmarshalPoint(c, id, out, i * pb, out[n * pb + i] === 1));
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

module.exports = { CURVE, init, register, pippenger, multiplyUnsafeBatch, multiplyBaseBatch, ed25519VerifyBatch, native };
