'use strict';
// Smoke test of the addon + shim.  The reference itself cannot be loaded on this Node (v12: no
// TypeScript), so a minimal Point class with the reference's CurvePointCons surface stands in;
// expected values come from the reference's fixtures (tests/golden, extracted from
// test/vectors/secp256k1/privates-2.txt) - k*G for k in the file.
const assert = require('assert');
const fs = require('fs');
const path = require('path');
const gpu = require('./noble_gpu.js');

const N = 0xfffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364141n;
const P = 0xfffffffffffffffffffffffffffffffffffffffffffffffffffffffefffffc2fn;
class Point {                                   // shape of weierstrass.ts:685-718
  constructor(x, y, inf) { this.x = x; this.y = y; this.inf = !!inf; }
  static fromAffine(a) { return (a.x === 0n && a.y === 0n) ? Point.ZERO : new Point(a.x, a.y); }
  toAffine() { return this.inf ? { x: 0n, y: 0n } : { x: this.x, y: this.y }; }
  negate() { return this.inf ? this : new Point(this.x, (P - this.y) % P); }
}
Point.ZERO = new Point(0n, 0n, true);
Point.BASE = new Point(0x79be667ef9dcbbac55a06295ce870b07029bfcdb2dce28d959f2815b16f81798n,
                       0x483ada7726a3c4655da4fbfc0e1108a8fd17b448a68554199c47d08ffb10d4b8n);
Point.Fp = { ORDER: P, BYTES: 32 };
Point.Fn = { ORDER: N, BYTES: 32 };
gpu.register(Point, gpu.CURVE.SECP256K1);

// validation happens before crossing, with the reference's messages
assert.throws(() => gpu.pippenger(Point, [Point.BASE, 5], [1n, 2n]), /invalid point at index 1/);
assert.throws(() => gpu.pippenger(Point, [Point.BASE], [N]), /invalid scalar at index 0/);
assert.throws(() => gpu.pippenger(Point, [Point.BASE], [1n, 2n]), /arrays of points and scalars must have equal length/);
assert.strictEqual(gpu.pippenger(Point, [], []), Point.ZERO);
assert.throws(() => gpu.multiplyUnsafeBatch(Point, [Point.BASE], [N]), /invalid scalar: out of range/);
console.log('validation OK, native:', gpu.native.version());

let haveGpu = true;
try { gpu.init(0); } catch (e) { haveGpu = false; console.log('no GPU here:', e.message); }
if (haveGpu) {
  const rows = JSON.parse(fs.readFileSync(path.join(__dirname, '..', 'tests', 'golden', 'secp256k1_privates2.json')));
  const ks = rows.map((r) => BigInt(r[0]));
  const exp = rows.map((r) => [BigInt('0x' + r[1]), BigInt('0x' + r[2])]);
  const viaBase = gpu.multiplyBaseBatch(Point, ks);
  const viaVar = gpu.multiplyUnsafeBatch(Point, ks.map(() => Point.BASE), ks);
  viaBase.forEach((p, i) => { assert.strictEqual(p.x, exp[i][0]); assert.strictEqual(p.y, exp[i][1]); });
  viaVar.forEach((p, i) => { assert.strictEqual(p.x, exp[i][0]); assert.strictEqual(p.y, exp[i][1]); });
  // MSM: sum_i s_i * (k_i G) == (sum s_i k_i mod n) G, checked through a single multiply
  const ss = ks.map((_, i) => BigInt(i * 7919 + 1));
  const tot = ks.reduce((a, k, i) => (a + k * ss[i]) % N, 0n);
  const msm = gpu.pippenger(Point, viaBase, ss);
  const ref = gpu.multiplyBaseBatch(Point, [tot])[0];
  assert.strictEqual(msm.x, ref.x); assert.strictEqual(msm.y, ref.y);
  assert.strictEqual(gpu.pippenger(Point, [Point.BASE, Point.BASE.negate(), Point.ZERO], [5n, 5n, 9n]), Point.ZERO);
  console.log('GPU smoke OK: multiplyBaseBatch / multiplyUnsafeBatch / pippenger match the reference vectors');
}
