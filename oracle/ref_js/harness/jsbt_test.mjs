// Stand-in for the test RUNNER the reference's test files import ('@paulmillr/jsbt/test.js': describe / it / should), which is
// not installed in this image.  TEST INFRASTRUCTURE (oracle/): it only collects and runs the reference's own test bodies; no
// arithmetic lives here.  Differences from the real runner: `it.runWhen(url)` never starts a run by itself - the driver script
// calls `runCollected({ filter })` after importing the test module, so that it can install the GPU backend first and count.
const root = { name: '', suites: [], tests: [], parent: null };
let cur = root;
export function describe(name, fn) {
  const s = { name, suites: [], tests: [], parent: cur };
  cur.suites.push(s);
  const prev = cur;
  cur = s;
  try { fn(); } finally { cur = prev; }
}
describe.skip = () => {};
describe.only = describe;
export function it(name, fn) { cur.tests.push({ name, fn, skip: false }); }
it.skip = (name, fn) => { cur.tests.push({ name, fn, skip: true }); };
it.only = it;
it.runWhen = () => {};
it.run = () => runCollected({});
it.runParallel = it.run;
export const should = it;
export const beforeEach = () => {};
export const afterEach = () => {};
function fullName(s, t) {
  const parts = [t];
  for (let x = s; x && x.parent; x = x.parent) parts.unshift(x.name);
  return parts.join(' > ');
}
// runs every collected test whose full name ("suite > suite > test") passes `filter` (RegExp or function); returns the tally
export async function runCollected(opts) {
  const filter = opts && opts.filter;
  const keep = (n) => (!filter ? true : typeof filter === 'function' ? filter(n) : filter.test(n));
  const res = { passed: 0, failed: 0, skipped: 0, filtered: 0, failures: [], names: [] };
  async function walk(s) {
    for (const t of s.tests) {
      const n = fullName(s, t.name);
      if (!keep(n)) { res.filtered++; continue; }
      if (t.skip) { res.skipped++; continue; }
      const t0 = Date.now();
      try {
        await t.fn();
        res.passed++;
        res.names.push(n);
        if (opts && opts.verbose) console.log('  ok   ' + n + ' (' + (Date.now() - t0) + ' ms)');
      } catch (e) {
        res.failed++;
        res.failures.push({ name: n, message: String(e && e.stack ? e.stack : e).slice(0, 1500) });
        console.log('  FAIL ' + n + ': ' + (e && e.message));
      }
    }
    for (const c of s.suites) await walk(c);
  }
  await walk(root);
  return res;
}
export function resetCollected() { root.suites = []; root.tests = []; cur = root; }
