// '@paulmillr/jsbt/benchmark-compare.js' (default export `compare(title, cases)`): see jsbt_bench.mjs.  TEST INFRASTRUCTURE.
import { compare } from './jsbt_bench.mjs';
export default compare;
