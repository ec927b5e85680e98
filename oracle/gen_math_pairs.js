'use strict';
/*
 * gen_math_pairs.js -- TEST INFRASTRUCTURE ONLY.
 * Writes tests/golden/v8_math_pairs.bin: N records of (x, Math.exp(x), Math.log(|x|))
 * as little-endian f64, produced by THIS Node's V8 -- the arithmetic the reference
 * actually runs on.  Pins oracle_math.h and csrc/amwg_math.h bit for bit.
 */
const fs = require('fs'), path = require('path');
const N = 120000, buf = Buffer.alloc(N * 24);
let s = 12345;
function rnd() { s = (Math.imul(s, 1103515245) + 12345) >>> 0; return s / 4294967296; }
for (let i = 0; i < N; i++) {
  let x; const m = i % 4;
  if (m === 0) x = (rnd() - 0.5) * 40; else if (m === 1) x = (rnd() - 0.5) * 1500;
  else if (m === 2) x = (rnd() - 0.5) * 2e-3; else x = Math.exp((rnd() - 0.5) * 200) * (rnd() < 0.5 ? -1 : 1);
  if (i < 8) x = [0, -0, 1, -1, 709.782712893384, -745.1332191019412, 1e-300, Infinity][i];
  buf.writeDoubleLE(x, i * 24); buf.writeDoubleLE(Math.exp(x), i * 24 + 8); buf.writeDoubleLE(Math.log(Math.abs(x)), i * 24 + 16);
}
fs.writeFileSync(path.join(__dirname, '..', 'tests', 'golden', 'v8_math_pairs.bin'), buf);
