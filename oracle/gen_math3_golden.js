'use strict';
/*
 * gen_math3_golden.js -- TEST INFRASTRUCTURE ONLY.
 * The trigonometric / hyperbolic / root functions of THIS Node's V8 (the arithmetic the reference runs on), for the twins in
 * csrc/amwg_math.h:
 *   tests/golden/v8_math3_pairs.bin  records of 15 f64: x, u, w, sin x, cos x, tan x, sinh x, cosh x, asinh x, cbrt x, log2 |x|,
 *                                    asin u, acos u, atanh u, acosh w          (u in [-1.05, 1.05], w >= 0.95)
 *   tests/golden/v8_atan2_pairs.bin  records of 3 f64: y, x, atan2(y, x)
 */
const fs = require('fs'), path = require('path');
const OUT = path.join(__dirname, '..', 'tests', 'golden');
let s = 987654321;
function rnd() { s = (Math.imul(s, 1103515245) + 12345) >>> 0; return s / 4294967296; }
{
  const N = 24000, W = 15, buf = Buffer.alloc(N * W * 8);
  const spx = [0, -0, 1, -1, 0.5, -0.5, Math.PI, -Math.PI, Math.PI / 2, Math.PI / 4, 3 * Math.PI / 4, 1e-9, -1e-9, 1e-300, Infinity, -Infinity, NaN, 22, -22, 710, -710, 710.4758600739439, 711, 1e22, -1e22, 1e300,
    0.6744, 0.67, 0.7853981633974483, 2.356194490192345, 1.5707963267948966, 102943.7, 1647099.3, 1647100, 3294198.6, 2, 8, 27, -27, 1e-310, 5e-324, 0.3465, 0.34657359027997264, 1.0397207708399179, 268435456, 268435457];
  const spu = [0, -0, 1, -1, 0.5, -0.5, 0.975, -0.975, 0.9749999, 1e-9, 1e-20, 0.4999999, 0.95, 1.0000001, -1.0000001, NaN];
  const spw = [1, 1.0000001, 2, 2.0000001, 1e9, 268435456, 268435457, 0.5, Infinity, NaN, 1e300];
  for (let i = 0; i < N; i++) {
    let x; const m = i % 8;
    if (i < spx.length) x = spx[i];
    else if (m === 0) x = (rnd() - 0.5) * 8; else if (m === 1) x = (rnd() - 0.5) * 2e-3; else if (m === 2) x = (rnd() - 0.5) * 1500;
    else if (m === 3) x = Math.exp((rnd() - 0.5) * 80) * (rnd() < 0.5 ? -1 : 1); else if (m === 4) x = (rnd() - 0.5) * 4e6; else if (m === 5) x = (rnd() - 0.5) * 60;
    else if (m === 6) x = Math.exp(rnd() * 700) * (rnd() < 0.5 ? -1 : 1); else x = (Math.floor(rnd() * 64) - 32) * (Math.PI / 2) + (rnd() - 0.5) * 1e-6;
    let u = i < spu.length ? spu[i] : (i % 3 === 0 ? (rnd() - 0.5) * 2.1 : (i % 3 === 1 ? (rnd() < 0.5 ? -1 : 1) * (1 - Math.exp(-rnd() * 30)) : (rnd() - 0.5) * Math.exp(-rnd() * 40)));
    let w = i < spw.length ? spw[i] : (i % 3 === 0 ? 0.95 + rnd() * 3 : (i % 3 === 1 ? 1 + Math.exp(-rnd() * 40) : Math.exp(rnd() * 700)));
    const v = [x, u, w, Math.sin(x), Math.cos(x), Math.tan(x), Math.sinh(x), Math.cosh(x), Math.asinh(x), Math.cbrt(x), Math.log2(Math.abs(x)), Math.asin(u), Math.acos(u), Math.atanh(u), Math.acosh(w)];
    v.forEach((q, j) => buf.writeDoubleLE(q, (i * W + j) * 8));
  }
  fs.writeFileSync(path.join(OUT, 'v8_math3_pairs.bin'), buf);
}
{
  const N = 24000, buf = Buffer.alloc(N * 24);
  const sp = [[0, 0], [-0, 0], [0, -0], [-0, -0], [1, 0], [-1, 0], [0, 1], [0, -1], [1, 1], [-1, -1], [Infinity, Infinity], [-Infinity, Infinity], [Infinity, -Infinity], [-Infinity, -Infinity], [1, Infinity], [1, -Infinity], [-1, -Infinity],
    [Infinity, 1], [-Infinity, 1], [NaN, 1], [1, NaN], [1e300, 1e-300], [1e300, -1e-300], [1e-300, 1e300], [1e-300, -1e300], [-1e-300, -1e300], [3, 1], [5, -2], [1e19, 1], [1e19, -1], [1, 1e19], [1, -1e19]];
  for (let i = 0; i < N; i++) {
    let y, x; const m = i % 4;
    if (i < sp.length) { y = sp[i][0]; x = sp[i][1]; }
    else if (m === 0) { y = (rnd() - 0.5) * 10; x = (rnd() - 0.5) * 10; }
    else if (m === 1) { y = Math.exp((rnd() - 0.5) * 100) * (rnd() < 0.5 ? -1 : 1); x = Math.exp((rnd() - 0.5) * 100) * (rnd() < 0.5 ? -1 : 1); }
    else if (m === 2) { y = (rnd() - 0.5) * 2; x = 1 + (rnd() - 0.5) * 1e-3; }
    else { y = Math.exp((rnd() - 0.5) * 1400) * (rnd() < 0.5 ? -1 : 1); x = Math.exp((rnd() - 0.5) * 1400) * (rnd() < 0.5 ? -1 : 1); }
    buf.writeDoubleLE(y, i * 24); buf.writeDoubleLE(x, i * 24 + 8); buf.writeDoubleLE(Math.atan2(y, x), i * 24 + 16);
  }
  fs.writeFileSync(path.join(OUT, 'v8_atan2_pairs.bin'), buf);
}
{
  // Math.hypot: records of 5 f64: a, b, c, hypot(a, b), hypot(a, b, c)
  const N = 12000, buf = Buffer.alloc(N * 40);
  const sp = [[0, 0, 0], [-0, 0, -0], [3, 4, 0], [3, 4, 12], [Infinity, NaN, 1], [NaN, 1, 2], [1, NaN, Infinity], [1e300, 1e300, 1e300], [1e-300, 1e-300, 1e-300], [5e-324, 0, 0], [1, 1e-20, 1e-20], [-Infinity, 1, 1], [1e308, 1e308, 0]];
  for (let i = 0; i < N; i++) {
    let v;
    if (i < sp.length) v = sp[i];
    else if (i % 3 === 0) v = [(rnd() - 0.5) * 20, (rnd() - 0.5) * 20, (rnd() - 0.5) * 20];
    else if (i % 3 === 1) v = [0, 1, 2].map(() => Math.exp((rnd() - 0.5) * 1400) * (rnd() < 0.5 ? -1 : 1));
    else v = [0, 1, 2].map(() => Math.exp((rnd() - 0.5) * 40) * (rnd() < 0.5 ? -1 : 1));
    [v[0], v[1], v[2], Math.hypot(v[0], v[1]), Math.hypot(v[0], v[1], v[2])].forEach((q, j) => buf.writeDoubleLE(q, i * 40 + j * 8));
  }
  fs.writeFileSync(path.join(OUT, 'v8_hypot_pairs.bin'), buf);
}
console.log('v8_math3_pairs.bin, v8_atan2_pairs.bin, v8_hypot_pairs.bin written');
