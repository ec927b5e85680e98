'use strict';
/*
 * gen_softplus_pairs.js -- TEST INFRASTRUCTURE ONLY.
 * Writes tests/golden/v8_softplus_pairs.bin: N records of (x, Math.log1p(Math.exp(x))) as little-endian f64, produced by THIS Node's V8
 * -- the arithmetic the reference runs when a closure writes a logistic likelihood as y*eta - Math.log1p(Math.exp(eta)).  Pins
 * csrc/amwg_math.h log1p_exp_v8 (tests/host/softplus_fuzz.cpp on the host, tests/test_gpu_math.py on the device) bit for bit.
 * Arguments: the range a logit link produces, the whole range of exp, and x = log(t) for t next to the thresholds of fdlibm's log1p
 * (sqrt(2) - 1; 1 + t next to a power of two or to sqrt(2) 2^k; 2^-29; 2^53).
 */
const fs = require('fs'), path = require('path');
const N = 100000, buf = Buffer.alloc(N * 16);
let s = 4711;
function rnd() { s = (Math.imul(s, 1103515245) + 12345) >>> 0; return s / 4294967296; }
const f64 = new Float64Array(1), u32 = new Uint32Array(f64.buffer);
function fromWords(hi, lo) { u32[1] = hi >>> 0; u32[0] = lo >>> 0; return f64[0]; }
const vw = [0x3FDA827A, 0x3e200000, 0x43400000, 0x3ff00000], mw = [0x3ff6a09e, 0x3ff00000, 0x3ffffffd, 0x3ff00004];
for (let i = 0; i < N; i++) {
  let x; const m = i % 5;
  if (m === 0) x = (rnd() - 0.5) * 16;
  else if (m === 1) x = rnd() * 60 - 22;
  else if (m === 2) x = (rnd() - 0.5) * 1500;
  else if (m === 3) x = Math.log(fromWords(vw[(i >> 3) % vw.length] + ((i >> 5) % 7) - 3, rnd() * 4294967296));
  else { const t = fromWords(mw[(i >> 3) % mw.length] + ((i >> 5) % 7) - 3, rnd() * 4294967296) * Math.pow(2, Math.floor(rnd() * 53)) - 1; x = t > 0 ? Math.log(t) : rnd(); }
  if (i < 10) x = [0, -0, 1, -1, -20, 36, 709.782712893384, -745.1332191019412, Infinity, -Infinity][i];
  buf.writeDoubleLE(x, i * 16); buf.writeDoubleLE(Math.log1p(Math.exp(x)), i * 16 + 8);
}
fs.writeFileSync(path.join(__dirname, '..', 'tests', 'golden', 'v8_softplus_pairs.bin'), buf);
