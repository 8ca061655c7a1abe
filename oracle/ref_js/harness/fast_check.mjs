// Stand-in for the property-testing library the reference's tests import ('fast-check'), which is not installed in this image.
// TEST INFRASTRUCTURE (oracle/): seeded generators for the handful of arbitraries test/point.test.ts and test/ed25519.test.ts
// use (integer, bigInt, string with a unit, array, tuple, map) and `assert(property(...))`; no shrinking - a failing run reports
// the generated arguments.  The seed is fixed, so a run is reproducible.
let state = 0x9e3779b97f4a7c15n;
function next64() {   // xorshift64, the generator of the reference's own makeRng (test/point.test.ts:536-559)
  state ^= BigInt.asUintN(64, state << 13n);
  state ^= state >> 7n;
  state ^= BigInt.asUintN(64, state << 17n);
  return state;
}
function belowBig(n) {            // uniform enough for tests: 64 bits of slack
  let bits = n.toString(2).length + 64, v = 0n;
  for (let i = 0; i < bits; i += 64) v = (v << 64n) | next64();
  return v % n;
}
function below(n) { return Number(belowBig(BigInt(n))); }
class Arb {
  constructor(gen) { this.gen = gen; }
  map(f) { return new Arb(() => f(this.gen())); }
  filter(f) { return new Arb(() => { for (;;) { const v = this.gen(); if (f(v)) return v; } }); }
  chain(f) { return new Arb(() => f(this.gen()).gen()); }
}
const range = (a, b, dmin, dmax) => {
  let min, max;
  if (a !== null && typeof a === 'object') { min = a.min === undefined ? dmin : a.min; max = a.max === undefined ? dmax : a.max; }
  else { min = a === undefined ? dmin : a; max = b === undefined ? dmax : b; }
  return [min, max];
};
export function integer(a, b) {
  const [min, max] = range(a, b, -0x80000000, 0x7fffffff);
  return new Arb(() => min + below(max - min + 1));
}
export const nat = (max) => integer(0, max === undefined ? 0x7fffffff : max);
export function bigInt(a, b) {
  const [min, max] = range(a, b, -(1n << 255n), 1n << 255n);
  // edge-biased like the real library: the ends of the range come up often
  return new Arb(() => { const r = below(16); return r === 0 ? min : r === 1 ? max : min + belowBig(max - min + 1n); });
}
export const bigUint = (max) => bigInt(0n, max === undefined ? (1n << 256n) - 1n : max);
export const bigUintN = (n) => bigInt(0n, (1n << BigInt(n)) - 1n);
export const boolean = () => new Arb(() => below(2) === 1);
export const constant = (v) => new Arb(() => v);
export const constantFrom = (...vs) => new Arb(() => vs[below(vs.length)]);
export const oneof = (...arbs) => new Arb(() => arbs[below(arbs.length)].gen());
export function array(arb, c) {
  const minL = (c && c.minLength) || 0, maxL = c && c.maxLength !== undefined ? c.maxLength : Math.max(minL, 10);
  return new Arb(() => { const n = minL + below(maxL - minL + 1); const out = []; for (let i = 0; i < n; i++) out.push(arb.gen()); return out; });
}
export function uint8Array(c) { return array(integer(0, 255), c).map((a) => Uint8Array.from(a)); }
export function string(c) {
  const unit = c && c.unit instanceof Arb ? c.unit : integer(0x20, 0x7e).map((n) => String.fromCharCode(n));
  return array(unit, c).map((a) => a.join(''));
}
export const hexaString = (c) => string({ ...(c || {}), unit: integer(0, 15).map((n) => '0123456789abcdef'[n]) });
export const tuple = (...arbs) => new Arb(() => arbs.map((a) => a.gen()));
export function record(shape) { return new Arb(() => { const o = {}; for (const k of Object.keys(shape)) o[k] = shape[k].gen(); return o; }); }
export function property(...args) {
  const fn = args.pop();
  return { arbs: args, fn };
}
export const asyncProperty = property;
export function assert(prop, opts) {
  const runs = (opts && opts.numRuns) || 100;
  if (opts && opts.seed !== undefined) state = BigInt.asUintN(64, BigInt(opts.seed)) || 1n;
  let pending = null;
  for (let i = 0; i < runs; i++) {
    const vals = prop.arbs.map((a) => a.gen());
    let r;
    try { r = prop.fn(...vals); } catch (e) {
      e.message = 'Property failed after ' + (i + 1) + ' runs on ' + show(vals) + ': ' + e.message;
      throw e;
    }
    if (r && typeof r.then === 'function') {   // async property: chain the remaining runs
      pending = (pending || Promise.resolve()).then(() => r);
    } else if (r === false) throw new Error('Property failed after ' + (i + 1) + ' runs on ' + show(vals));
  }
  return pending || undefined;
}
function show(v) { return JSON.stringify(v, (k, x) => (typeof x === 'bigint' ? x.toString() + 'n' : x)).slice(0, 400); }
export default { integer, nat, bigInt, bigUint, bigUintN, boolean, constant, constantFrom, oneof, array, uint8Array, string, hexaString, tuple, record, property, asyncProperty, assert };
