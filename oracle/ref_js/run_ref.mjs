// Runs the REFERENCE's own code (paulmillr/noble-curves, type-stripped into oracle/_ref/refjs.bundle by downlevel.py) on this
// machine's Node: timings for bench.py's cpu_baseline (`kind: "reference"`) and known answers that pin the Python oracle
// (tests/test_reference_js.py).  TEST INFRASTRUCTURE - the product never runs this.
//
//   node oracle/ref_js/run_ref.mjs <command> <in.bin> <out.bin> [args...]     -> one JSON line on stdout
// Wire formats are those of include/ncg.h: little-endian canonical residues, affine x || y, infinity (0,0) / Edwards (0,1),
// scalars 32 B LE.
//   mul_unsafe  <curve> in: n x (point || scalar)   out: n x point      Point.multiplyUnsafe (weierstrass.ts:915-928, edwards.ts:568-577)
//   mul         <curve> same                                             Point.multiply (weierstrass.ts:900-907); k = 0 rows are skipped (it throws)
//   pippenger   <curve> in: n x point, then n x scalar   out: 1 point    pippenger (abstract/curve.ts:863-905)
//   ed25519_verify in: n x (sig 64 || pk 32 || msglen u32 || msg padded to 64)  out: n bytes   ed25519.verify (edwards.ts:942-989), zip215 flag in args
//   h2c <g1|g2> <dstHex> [encode]  in: n x (msglen u32 || msg padded to 64)   out: n x point   hashToCurve / encodeToCurve (hash-to-curve.ts:487-500)
//   fft <direct|inverse> <brpIn 0/1> <brpOut 0/1>   in: N x 32 B LE residues of Fr    out: N x 32 B   FFT(rootsOfUnity(Fr), Fr) (fft.ts:518-577)
//   codec <curve>  in: n x point   out: n x (compressed bytes: 33 / 32 / 48 / 96)      Point.toBytes(true) -> Point.fromBytes -> must equal the input
//   point_bench <seconds>   benchmark/point.ts:20-32 on secp256k1: Point_mul / Point_mulUns of one point by 2^180 - 15820, and by random scalars
import './polyfill.mjs';
import fs from 'fs';
import { pippenger } from './abstract/curve.mjs';
import { FFT, rootsOfUnity } from './abstract/fft.mjs';
import { bls12_381 } from './bls12-381.mjs';
import { ed25519 } from './ed25519.mjs';
import { secp256k1 } from './secp256k1.mjs';

const now = () => Number(process.hrtime.bigint()) / 1e6;
const le2n = (b, o, len) => {
  let v = 0n;
  for (let i = len - 1; i >= 0; i--) v = (v << 8n) | BigInt(b[o + i]);
  return v;
};
const n2le = (v, b, o, len) => {
  for (let i = 0; i < len; i++) {
    b[o + i] = Number(v & 0xffn);
    v >>= 8n;
  }
};
const CURVES = {
  secp256k1: { P: secp256k1.Point, fb: 32, fp2: false },
  ed25519: { P: ed25519.Point, fb: 32, fp2: false, edwards: true },
  bls12_381_g1: { P: bls12_381.G1.Point, fb: 48, fp2: false },
  bls12_381_g2: { P: bls12_381.G2.Point, fb: 48, fp2: true },
};
function readPoint(C, b, o) {
  const fb = C.fb;
  if (C.fp2) {
    const x = { c0: le2n(b, o, fb), c1: le2n(b, o + fb, fb) }, y = { c0: le2n(b, o + 2 * fb, fb), c1: le2n(b, o + 3 * fb, fb) };
    if (x.c0 === 0n && x.c1 === 0n && y.c0 === 0n && y.c1 === 0n) return C.P.ZERO;
    return C.P.fromAffine({ x, y });
  }
  const x = le2n(b, o, fb), y = le2n(b, o + fb, fb);
  if (C.edwards ? x === 0n && y === 1n : x === 0n && y === 0n) return C.P.ZERO;
  return C.P.fromAffine({ x, y });
}
function writePoint(C, p, b, o) {
  const fb = C.fb;
  const z = p.is0();
  const a = z ? null : p.toAffine();
  if (C.fp2) {
    n2le(z ? 0n : a.x.c0, b, o, fb);
    n2le(z ? 0n : a.x.c1, b, o + fb, fb);
    n2le(z ? 0n : a.y.c0, b, o + 2 * fb, fb);
    n2le(z ? 0n : a.y.c1, b, o + 3 * fb, fb);
  } else {
    n2le(z ? 0n : a.x, b, o, fb);
    n2le(z ? (C.edwards ? 1n : 0n) : a.y, b, o + fb, fb);
  }
}
const pb = (C) => (C.fp2 ? 4 : 2) * C.fb;

const [cmd, inFile, outFile, ...args] = process.argv.slice(2);
const res = { cmd, node: process.version };
if (cmd === 'mul_unsafe' || cmd === 'mul') {
  const C = CURVES[args[0]];
  const buf = fs.readFileSync(inFile);
  const rec = pb(C) + 32, n = buf.length / rec;
  const pts = [], ks = [];
  for (let i = 0; i < n; i++) {
    pts.push(readPoint(C, buf, i * rec));
    ks.push(le2n(buf, i * rec + pb(C), 32));
  }
  const out = new Uint8Array(n * pb(C));
  const outs = new Array(n);
  const t0 = now();
  for (let i = 0; i < n; i++) outs[i] = cmd === 'mul' ? (ks[i] === 0n ? C.P.ZERO : pts[i].multiply(ks[i])) : pts[i].multiplyUnsafe(ks[i]);
  res.ms = now() - t0;
  for (let i = 0; i < n; i++) writePoint(C, outs[i], out, i * pb(C));
  fs.writeFileSync(outFile, out);
  res.n = n;
  res.per_s = n / (res.ms / 1e3);
} else if (cmd === 'pippenger') {
  const C = CURVES[args[0]];
  const buf = fs.readFileSync(inFile);
  const n = buf.length / (pb(C) + 32);
  const pts = [], ks = [];
  for (let i = 0; i < n; i++) pts.push(readPoint(C, buf, i * pb(C)));
  for (let i = 0; i < n; i++) ks.push(le2n(buf, n * pb(C) + i * 32, 32));
  const t0 = now();
  const r = pippenger(C.P, pts, ks);
  res.ms = now() - t0;
  const out = new Uint8Array(pb(C));
  writePoint(C, r, out, 0);
  fs.writeFileSync(outFile, out);
  res.n = n;
  res.per_s = n / (res.ms / 1e3);
} else if (cmd === 'ed25519_verify') {
  const zip215 = args[0] !== 'strict';
  const buf = fs.readFileSync(inFile);
  const rec = 64 + 32 + 4 + 64, n = buf.length / rec;
  const out = new Uint8Array(n);
  const items = [];
  for (let i = 0; i < n; i++) {
    const o = i * rec;
    const len = buf.readUInt32LE(o + 96);
    items.push([new Uint8Array(buf.slice(o, o + 64)), new Uint8Array(buf.slice(o + 100, o + 100 + len)), new Uint8Array(buf.slice(o + 64, o + 96))]);
  }
  const t0 = now();
  for (let i = 0; i < n; i++) {
    let ok = false;
    try {
      ok = ed25519.verify(items[i][0], items[i][1], items[i][2], { zip215 });
    } catch (e) {
      ok = false;
    }
    out[i] = ok ? 1 : 0;
  }
  res.ms = now() - t0;
  fs.writeFileSync(outFile, out);
  res.n = n;
  res.per_s = n / (res.ms / 1e3);
} else if (cmd === 'h2c') {
  const H = args[0] === 'g2' ? bls12_381.G2 : bls12_381.G1;
  const C = CURVES[args[0] === 'g2' ? 'bls12_381_g2' : 'bls12_381_g1'];
  const DST = new Uint8Array(Buffer.from(args[1], 'hex'));
  const encode = args[2] === 'encode';
  const buf = fs.readFileSync(inFile);
  const rec = 4 + 64, n = buf.length / rec;
  const out = new Uint8Array(n * pb(C));
  const t0 = now();
  for (let i = 0; i < n; i++) {
    const len = buf.readUInt32LE(i * rec);
    const msg = new Uint8Array(buf.slice(i * rec + 4, i * rec + 4 + len));
    const P = encode ? H.encodeToCurve(msg, { DST }) : H.hashToCurve(msg, { DST });
    writePoint(C, P, out, i * pb(C));
  }
  res.ms = now() - t0;
  fs.writeFileSync(outFile, out);
  res.n = n;
  res.per_s = n / (res.ms / 1e3);
} else if (cmd === 'fft') {
  const Fr = bls12_381.fields.Fr;
  const f = FFT(rootsOfUnity(Fr), Fr);
  const buf = fs.readFileSync(inFile);
  const N = buf.length / 32;
  const vals = [];
  for (let i = 0; i < N; i++) vals.push(le2n(buf, i * 32, 32));
  const brpIn = args[1] === '1', brpOut = args[2] === '1';
  const t0 = now();
  const r = args[0] === 'inverse' ? f.inverse(vals, brpIn, brpOut) : f.direct(vals, brpIn, brpOut);
  res.ms = now() - t0;
  const out = new Uint8Array(N * 32);
  for (let i = 0; i < N; i++) n2le(r[i], out, i * 32, 32);
  fs.writeFileSync(outFile, out);
  res.n = N;
  res.per_s = N / (res.ms / 1e3);
} else if (cmd === 'codec') {
  const C = CURVES[args[0]];
  const buf = fs.readFileSync(inFile);
  const n = buf.length / pb(C);
  let out = null, L = 0;
  for (let i = 0; i < n; i++) {
    const P = readPoint(C, buf, i * pb(C));
    const enc = C.edwards ? P.toBytes() : P.toBytes(true);
    if (!out) {
      L = enc.length;
      out = new Uint8Array(n * L);
    }
    if (enc.length !== L) throw new Error('codec: encoding length changed');
    if (!C.P.fromBytes(enc).equals(P)) throw new Error('codec: fromBytes(toBytes(P)) != P at ' + i);
    out.set(enc, i * L);
  }
  fs.writeFileSync(outFile, out || new Uint8Array(0));
  res.n = n;
  res.enc_len = L;
} else if (cmd === 'point_bench') {
  // benchmark/point.ts:20-32: one public-key point, Point.multiply / multiplyUnsafe by the literal 2^180 - 15820; plus the
  // 1 000-random-scalar form BASELINE configs[0] names
  const seconds = parseFloat(inFile || '2');
  const P = secp256k1.Point;
  P.BASE.precompute(6, false);
  const point = P.fromBytes(secp256k1.getPublicKey(secp256k1.utils.randomSecretKey(), true));
  const scalar = 2n ** 180n - 15820n;
  const loop = (f) => {
    const t0 = now();
    let n = 0;
    while (now() - t0 < seconds * 1e3) {
      f(n);
      n++;
    }
    return n / ((now() - t0) / 1e3);
  };
  let s = 0x9e3779b97f4a7c15n;
  const rnd = () => {
    s ^= (s << 13n) & ((1n << 64n) - 1n);
    s ^= s >> 7n;
    s ^= (s << 17n) & ((1n << 64n) - 1n);
    return s;
  };
  const ks = [];
  for (let i = 0; i < 1000; i++) ks.push((((rnd() << 192n) | (rnd() << 128n) | (rnd() << 64n) | rnd()) % (P.Fn.ORDER - 1n)) + 1n);
  res.Point_mul = loop(() => point.multiply(scalar));
  res.Point_mulUns = loop(() => point.multiplyUnsafe(scalar));
  res.Point_mul_random = loop((i) => point.multiply(ks[i % 1000]));
  res.Point_mulUns_random = loop((i) => point.multiplyUnsafe(ks[i % 1000]));
  res.equal = point.multiply(scalar).equals(point.multiplyUnsafe(scalar));
} else {
  console.error('unknown command ' + cmd);
  process.exit(2);
}
console.log(JSON.stringify(res));
