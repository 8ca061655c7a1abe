'use strict';
// Smoke test of the addon + shim with NO reference on the path: a minimal Point class with the reference's
// CurvePointCons surface stands in (any class with that surface can be registered); expected values come from the
// reference's fixtures (tests/golden, extracted from test/vectors/secp256k1/privates-2.txt) - k*G for k in the file.
// The shim driven by the reference's OWN classes (secp256k1.Point, bls12_381.G1/G2.Point, ed25519.Point from the
// type-stripped bundle) and compared with the reference's own pippenger / multiply on the same objects is
// addon/ref_dropin_test.mjs.
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
Point.Fn = { ORDER: N, BYTES: 32, BITS: 256 };
gpu.register(Point, gpu.CURVE.SECP256K1);

// validation happens before crossing, with the reference's messages
assert.throws(() => gpu.pippenger(Point, [Point.BASE, 5], [1n, 2n]), /invalid point at index 1/);
assert.throws(() => gpu.pippenger(Point, [Point.BASE], [N]), /invalid scalar at index 0/);
assert.throws(() => gpu.pippenger(Point, [Point.BASE], [1n, 2n]), /arrays of points and scalars must have equal length/);
assert.strictEqual(gpu.pippenger(Point, [], []), Point.ZERO);
assert.throws(() => gpu.multiplyUnsafeBatch(Point, [Point.BASE], [N]), /invalid scalar: out of range/);
console.log('validation OK, native:', gpu.native.version());

let haveGpu = true;
// the device-set context with one device: pippenger goes through ncg_msm_multi, everything else through its first context
try { gpu.initMulti([0]); } catch (e) { haveGpu = false; console.log('no GPU here:', e.message); }
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
  {  // packed columns in place of Point[] / BigInt[]: same result, same scalar rule
    const pp = gpu.packPoints(Point, viaBase), ps = gpu.packScalars(ss, Point.Fn);
    const m2 = gpu.pippenger(Point, pp, ps), m3 = gpu.pippenger(Point, pp, new BigUint64Array(ps.buffer, ps.byteOffset, ps.length / 8));
    assert.strictEqual(m2.x, msm.x); assert.strictEqual(m2.y, msm.y); assert.strictEqual(m3.x, msm.x); assert.strictEqual(m3.y, msm.y);
    const badS = Uint8Array.from(ps); badS.fill(0xff, 0, 32);
    assert.throws(() => gpu.pippenger(Point, pp, badS), /invalid scalar at index 0/);
    assert.throws(() => gpu.pippenger(Point, pp.subarray(0, pp.length - 64), ps), /arrays of points and scalars must have equal length/);
  }
  const ref = gpu.multiplyBaseBatch(Point, [tot])[0];
  assert.strictEqual(msm.x, ref.x); assert.strictEqual(msm.y, ref.y);
  assert.strictEqual(gpu.pippenger(Point, [Point.BASE, Point.BASE.negate(), Point.ZERO], [5n, 5n, 9n]), Point.ZERO);
  // codecs: SEC1 round trip of the k*G vectors; ZERO is rejected like the reference
  const enc = gpu.toBytesBatch(Point, viaBase);
  enc.forEach((e, i) => { assert.strictEqual(e.length, 33); assert.strictEqual(e[0], 2 + Number(exp[i][1] & 1n)); });
  gpu.fromBytesBatch(Point, enc).forEach((p, i) => { assert.strictEqual(p.x, exp[i][0]); assert.strictEqual(p.y, exp[i][1]); });
  const bad = Uint8Array.from(enc[0]); bad[0] = 5;
  assert.strictEqual(gpu.fromBytesBatch(Point, [bad])[0], null);
  assert.throws(() => gpu.toBytesBatch(Point, [Point.ZERO]), /bad point: ZERO/);
  const agg = gpu.aggregateFromBytes(Point, enc);                 // sum of k_i G == (sum k_i) G
  const aggRef = gpu.multiplyBaseBatch(Point, [ks.reduce((a, k) => (a + k) % N, 0n)])[0];
  assert.strictEqual(agg.x, aggRef.x); assert.strictEqual(agg.y, aggRef.y);
  assert.throws(() => gpu.aggregateFromBytes(Point, [enc[0], bad]), /invalid point encoding at index 1/);
  // FFT over Fr: the reference's 'Basic FFT' known answer (test/fft.test.ts:221-251) and round trips
  const kat = JSON.parse(fs.readFileSync(path.join(__dirname, '..', 'tests', 'golden', 'fft_kat.json')));
  const fin = kat.basic_input.map(BigInt), fexp = kat.basic_exp.map(BigInt);
  assert.deepStrictEqual(gpu.fftFr(fin), fexp);
  assert.deepStrictEqual(gpu.fftFr(gpu.fftFr(fin), { inverse: true }), fin);
  assert.deepStrictEqual(gpu.fftFr(gpu.fftFr(fin, { brpOutput: true }), { inverse: true, brpInput: true }), fin);
  assert.throws(() => gpu.fftFr([1n, 2n, 3n]), /FFT: Polynomial size should be power of two/);
  // hash-to-curve + multiply + encode: the reference's priv:msg:sig vectors (test/bls12-381.test.ts:953-966)
  const BP = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaabn;
  const BR = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001n;
  class G2 {                                    // Fp2 coordinates as { c0, c1 } like the reference (tower.ts)
    constructor(x, y, inf) { this.x = x; this.y = y; this.inf = !!inf; }
    static fromAffine(a) { return (a.x.c0 === 0n && a.x.c1 === 0n && a.y.c0 === 0n && a.y.c1 === 0n) ? G2.ZERO : new G2(a.x, a.y); }
    toAffine() { return this.inf ? { x: { c0: 0n, c1: 0n }, y: { c0: 0n, c1: 0n } } : { x: this.x, y: this.y }; }
  }
  G2.ZERO = new G2(null, null, true);
  G2.Fp = { ORDER: BP * BP, BYTES: 96 };
  G2.Fn = { ORDER: BR, BYTES: 32 };
  gpu.register(G2, gpu.CURVE.BLS12_381_G2);
  const sig = JSON.parse(fs.readFileSync(path.join(__dirname, '..', 'tests', 'golden', 'bls12_381_sig_vectors.json'))).g2.slice(0, 8);
  const H = gpu.hashToCurveBatch(G2, sig.map((r) => Buffer.from(r.msg, 'hex')));
  const S = gpu.multiplyUnsafeBatch(G2, H, sig.map((r) => BigInt('0x' + r.priv) % BR));
  gpu.toBytesBatch(G2, S).forEach((e, i) => assert.strictEqual(Buffer.from(e).toString('hex'), sig[i].sig));
  // a G2 set verified to lie in the subgroup takes the endomorphism MSM: same result as the generic path
  {
    const scG = sig.map((r, i) => (BigInt('0x' + r.priv) * BigInt(i + 3)) % BR);
    const plain = gpu.uploadPoints(G2, S), fast = gpu.uploadPoints(G2, S, { checkSubgroup: true });
    assert.strictEqual(plain.inSubgroup, false); assert.strictEqual(fast.inSubgroup, true);
    const a = gpu.pippengerResident(plain, scG), b = gpu.pippengerResident(fast, scG), c0 = gpu.pippenger(G2, S, scG);
    for (const r of [b, c0]) { assert.strictEqual(r.x.c0, a.x.c0); assert.strictEqual(r.x.c1, a.x.c1); assert.strictEqual(r.y.c0, a.y.c0); assert.strictEqual(r.y.c1, a.y.c1); }
    plain.free(); fast.free();
  }
  // resident point sets: upload once (points, or their encodings decoded on the device), then MSMs with only
  // the scalars crossing - as BigInt[] (validated) or as packed bytes
  const set = gpu.uploadPoints(Point, viaBase);
  const setEnc = gpu.uploadEncoded(Point, Uint8Array.from([].concat(...enc.map((e) => Array.from(e)))));
  for (const st of [set, setEnc]) {
    const r1 = gpu.pippengerResident(st, ss);
    assert.strictEqual(r1.x, ref.x); assert.strictEqual(r1.y, ref.y);
    const packed = new Uint8Array(32 * ss.length);
    ss.forEach((v, i) => { let t = v; for (let j = 0; j < 32; j++) { packed[32 * i + j] = Number(t & 0xffn); t >>= 8n; } });
    const r2 = gpu.pippengerResident(st, packed);
    assert.strictEqual(r2.x, ref.x); assert.strictEqual(r2.y, ref.y);
    const mv = gpu.multiplyUnsafeBatchResident(st, ss);
    const mvRef = gpu.multiplyUnsafeBatch(Point, viaBase, ss);
    mv.forEach((p, i) => { assert.strictEqual(p.x, mvRef[i].x); assert.strictEqual(p.y, mvRef[i].y); });
  }
  assert.throws(() => gpu.pippengerResident(set, ss.slice(1)), /arrays of points and scalars must have equal length/);
  assert.throws(() => gpu.pippengerResident(set, ss.map(() => N)), /invalid scalar at index 0/);
  {  // packed scalars get the same range rule: the group order itself (and 2^256 - 1) must be refused, on every resident entry point
    const bad1 = new Uint8Array(32 * ss.length), bad2 = new Uint8Array(32 * ss.length).fill(0);
    { let t = N; for (let j = 0; j < 32; j++) { bad1[32 + j] = Number(t & 0xffn); t >>= 8n; } }
    bad2.fill(0xff, 64, 96);
    assert.throws(() => gpu.pippengerResident(set, bad1), /invalid scalar at index 1/);
    assert.throws(() => gpu.multiplyUnsafeBatchResident(set, bad2), /invalid scalar at index 2/);
  }
  assert.throws(() => gpu.native.packBigInts([1n << 512n], 32), /out of range/);
  assert.throws(() => gpu.uploadEncoded(Point, Uint8Array.from(Array.from(enc[0]).concat(Array.from(bad)))), /invalid point encoding at index 1/);
  set.free(); setEnc.free();
  // interleavedMSMUnsafe (test/point.test.ts:309-317): 3G + 5*2G + 7*4G + 11*8G = 129G for every window size,
  // window 1 throws, fewer scalars than points = trailing zeros, more = error
  {
    const pts2 = gpu.multiplyBaseBatch(Point, [1n, 2n, 4n, 8n]);
    const want = gpu.multiplyBaseBatch(Point, [129n, 13n]);
    for (let W = 2; W <= 10; W++) {
      const mul = gpu.interleavedMSMUnsafe(Point, pts2, W);
      const r = mul([3n, 5n, 7n, 11n]);
      assert.strictEqual(r.x, want[0].x); assert.strictEqual(r.y, want[0].y);
      if (W === 4) {
        const short = mul([3n, 5n]);
        assert.strictEqual(short.x, want[1].x); assert.strictEqual(short.y, want[1].y);
        assert.throws(() => mul([1n, 1n, 1n, 1n, 1n]), /array of scalars must not be larger than array of points/);
        assert.throws(() => mul([N]), /invalid scalar at index 0/);
      }
    }
    assert.throws(() => gpu.interleavedMSMUnsafe(Point, pts2, 1), /window/);
    assert.strictEqual(gpu.interleavedMSMUnsafe(Point, [], 4)([]), Point.ZERO);
  }
  // timing of the resident path at 2^16 (SURVEY 8a gotcha 8: marshalling is the end-to-end cost)
  {
    const n = 1 << 16;
    const big = gpu.multiplyBaseBatch(Point, Array.from({ length: 64 }, (_, i) => BigInt(i + 1)));
    const pts = Array.from({ length: n }, (_, i) => big[i % 64]);
    const bs = gpu.uploadPoints(Point, pts);
    const scB = new Uint8Array(32 * n);
    for (let i = 0; i < n; i++) for (let j = 0; j < 31; j++) scB[32 * i + j] = (i * 131 + j * 17 + 7) & 0xff;
    gpu.pippengerResident(bs, scB);
    const t0 = process.hrtime.bigint();
    const reps = 5;
    for (let r = 0; r < reps; r++) gpu.pippengerResident(bs, scB);
    const ms = Number(process.hrtime.bigint() - t0) / 1e6 / reps;
    console.log('resident pippenger 2^16 (packed scalars): ' + ms.toFixed(2) + ' ms per call');
    bs.free();
  }
  // ECDSA verify: the reference's RFC 6979 vectors (test/vectors/secp256k1/ecdsa.json via tests/golden): keys from the GPU,
  // every signature verifies, a flipped bit does not
  {
    const vec = JSON.parse(fs.readFileSync(path.join(__dirname, '..', 'tests', 'golden', 'secp256k1_ecdsa.json'))).valid.slice(0, 24);
    const keys = gpu.toBytesBatch(Point, gpu.multiplyBaseBatch(Point, vec.map((v) => BigInt('0x' + v.d))));
    const hex = (h) => Uint8Array.from(Buffer.from(h, 'hex'));
    const items = vec.map((v, i) => ({ sig: hex(v.signature), msgHash: hex(v.m), publicKey: keys[i] }));
    assert.deepStrictEqual(gpu.ecdsaVerifyBatch(items), items.map(() => true));
    items[3].sig[40] ^= 1; items[5].msgHash[0] ^= 1; items[7].publicKey = keys[8];
    assert.deepStrictEqual(gpu.ecdsaVerifyBatch(items), items.map((_, i) => ![3, 5, 7].includes(i)));
    // recovery: exactly one recovery id gives the signer's key back
    const rec = [];
    vec.slice(0, 6).forEach((v) => { for (let r = 0; r < 4; r++) rec.push({ sig: Uint8Array.from([r, ...hex(v.signature)]), msgHash: hex(v.m) }); });
    const keysBack = gpu.ecdsaRecoverBatch(rec);
    for (let i = 0; i < 6; i++) {
      const hits = keysBack.slice(4 * i, 4 * i + 4).filter((k) => k !== null && Buffer.from(k).equals(Buffer.from(keys[i]))).length;
      assert.strictEqual(hits, 1);
    }
    // BIP-340 vectors (test/vectors/secp256k1/schnorr.csv via tests/golden): hash and verification on the device
    const sv = JSON.parse(fs.readFileSync(path.join(__dirname, '..', 'tests', 'golden', 'secp256k1_schnorr.json')));
    assert.deepStrictEqual(gpu.schnorrVerifyBatch(sv.map((r) => ({ sig: hex(r.sig), msg: hex(r.msg), publicKey: hex(r.pub) }))), sv.map((r) => r.result));
  }
  // ed25519 from messages (hash on the device): RFC 8032 test 2 and a corrupted copy
  {
    const hex = (h) => Uint8Array.from(Buffer.from(h, 'hex'));
    const pk = hex('3d4017c3e843895a92b70aa74d1b7ebc9c982ccf2ec4968cc0cd55f12af4660c');
    const msg = hex('72');
    const sg = hex('92a009a9f0d4cab8720e820b5f642540a2b27b5416503f8fb3762223ebdb69da085ac1e43e15996e458f3613d0f11d8c387b2eaeb4302aeeb00d291612bb0c00');
    const badSg = Uint8Array.from(sg); badSg[40] ^= 1;
    assert.deepStrictEqual(gpu.ed25519VerifyBatchDevice([{ sig: sg, msg, publicKey: pk }, { sig: badSg, msg, publicKey: pk },
                                                         { sig: sg, msg: hex('73'), publicKey: pk }]), [true, false, false]);
    assert.deepStrictEqual(gpu.ed25519VerifyBatch([{ sig: sg, msg, publicKey: pk }]), [true]);
  }
  console.log('GPU smoke OK: multiplyBaseBatch / multiplyUnsafeBatch / pippenger match the reference vectors');
  console.log('GPU smoke OK: codecs, FFT known answer and hash-to-curve signature vectors');
}
