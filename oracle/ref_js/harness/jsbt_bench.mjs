// Stand-in for the reference's benchmark helpers ('@paulmillr/jsbt/benchmark.js' default export + `section`, and
// '@paulmillr/jsbt/benchmark-compare.js'), which are not installed in this image.  TEST INFRASTRUCTURE (oracle/): each case runs a
// bounded number of times, the mean goes to stdout and into `results` for the driver script.
export const results = [];
let currentSection = '';
const budgetMs = Number(process.env.NCG_JSBT_BUDGET_MS || 300);
export function section(name) { currentSection = name; console.log('# ' + name); }
// NCG_JSBT_FILTER (a regular expression, read at call time): cases whose name does not match are recorded as skipped and NOT run -
// how the driver script runs lines 64-79 of benchmark/bls12-381.ts without the minutes of pairings around them
async function timeCase(name, samples, fn) {
  const f = process.env.NCG_JSBT_FILTER;
  if (f && !new RegExp(f).test(name)) {
    results.push({ section: currentSection, name, skipped: true });
    return null;
  }
  let n = 0;
  const t0 = process.hrtime.bigint();
  let el = 0;
  do {
    await fn();
    n++;
    el = Number(process.hrtime.bigint() - t0) / 1e6;
  } while (samples !== 'once' && n < (typeof samples === 'number' ? samples : 1000) && el < budgetMs);
  const rec = { section: currentSection, name, runs: n, ms_per_op: el / n };
  results.push(rec);
  console.log(name + ' x ' + (1000 / rec.ms_per_op).toFixed(rec.ms_per_op > 100 ? 3 : 0) + ' ops/sec @ ' + rec.ms_per_op.toFixed(3) + ' ms/op (' + n + ' runs)');
  return rec;
}
export default async function bench(name, a, b) {
  const fn = typeof a === 'function' ? a : b;
  const samples = typeof a === 'function' ? undefined : a;
  return timeCase(name, samples, fn);
}
export const mark = bench;
export async function compare(title, cases) {
  section(title);
  for (const [name, fn] of Object.entries(cases)) await timeCase(name, undefined, fn);
}
