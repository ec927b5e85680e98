'use strict';
/*
 * gen_mod_golden.js -- TEST INFRASTRUCTURE ONLY.
 * The `%` operator and ToInt32 (`x | 0`) of THIS Node's V8 -- the arithmetic an arbitrary reference closure may use
 * (/root/reference/README.md:122, mcmc.js:958-960) -- for js_mod / js_toint32 in csrc/amwg_user.h:
 *   tests/golden/v8_mod_pairs.bin   records of 4 f64: a, b, a % b, (a | 0)
 */
const fs = require('fs'), path = require('path');
const OUT = path.join(__dirname, '..', 'tests', 'golden');
let s = 20260926;
function rnd() { s = (Math.imul(s, 1103515245) + 12345) >>> 0; return s / 4294967296; }
function sgn() { return rnd() < 0.5 ? -1 : 1; }
const sp = [0, -0, 1, -1, 3, -3, 0.1, -0.1, 5.5, -5.5, 6, -6, 1e20, -1e20, 1e300, -1e300, 5e-324, -5e-324, 2.2250738585072014e-308, 1e-310, -1e-310,
  Infinity, -Infinity, NaN, 4294967296, -4294967296, 2147483648, -2147483648, 2147483647, -2147483649, 4294967295.5, 1.7976931348623157e308, 0.5, -0.5, 2, -2];
const N = 40000, buf = Buffer.alloc(N * 32);
let i = 0;
function put(a, b) { if (i >= N) return; [a, b, a % b, (a | 0)].forEach((q, j) => buf.writeDoubleLE(q, i * 32 + j * 8)); i++; }
for (const a of sp) for (const b of sp) put(a, b);                              // 1296 special pairs incl. every +-0 / inf / NaN combination
while (i < N) {
  const m = i % 8;
  let a, b;
  if (m === 0) { a = (rnd() - 0.5) * 200; b = (rnd() - 0.5) * 20; }
  else if (m === 1) { a = Math.round((rnd() - 0.5) * 2000); b = Math.round((rnd() - 0.5) * 40) || 3; }             // integers: many exact-zero results
  else if (m === 2) { a = Math.exp((rnd() - 0.5) * 1400) * sgn(); b = Math.exp((rnd() - 0.5) * 1400) * sgn(); }     // any exponent gap, both directions
  else if (m === 3) { b = Math.exp((rnd() - 0.5) * 60) * sgn(); a = b * Math.round(rnd() * 1e6) * sgn(); }           // (rounded) multiples of b
  else if (m === 4) { a = rnd() * 1e-307 * sgn(); b = rnd() * 1e-309 * sgn(); }                                    // subnormal divisors / results
  else if (m === 5) { a = Math.round((rnd() - 0.5) * 3e10); b = 4294967296 * (rnd() < 0.5 ? 1 : 0.5); }           // the ToInt32 range reduction
  else if (m === 6) { a = Math.exp(rnd() * 700) * sgn(); b = (rnd() + 0.01) * 10; }                                 // huge / small: thousands of quotient bits
  else { a = (rnd() - 0.5) * 10; b = a * (1 + (rnd() - 0.5) * 1e-15); }                                            // |a| ~ |b|
  put(a, b);
}
fs.writeFileSync(path.join(OUT, 'v8_mod_pairs.bin'), buf);
console.log('v8_mod_pairs.bin written');
