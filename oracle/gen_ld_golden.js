'use strict';
/*
 * gen_ld_golden.js -- TEST INFRASTRUCTURE ONLY.
 * Writes tests/golden/ld_values.bin (records of 6 little-endian f64: id, x, a, b, c, value) by
 * calling the UNMODIFIED reference's distributions.js (/root/reference, build container only) on
 * seeded arguments, and tests/golden/v8_pow_pairs.bin (x, y, Math.pow(x, y)) from this Node's V8.
 * ids: 0 norm 1 unif 2 beta 3 bern 4 pois 5 cauchy 6 laplace 7 gamma 8 invgamma 9 lnorm 10 pareto
 *      11 t 12 weibull 13 logis 14 exp 15 binom 16 nbinom 17 hyper 18 lgamma 19 lfactorial 20 lchoose 21 lbeta
 */
const fs = require('fs'), path = require('path');
const ld = require(path.join(require('./ref_dir.js').refDir() || '/root/reference', 'distributions.js'));
const OUT = path.join(__dirname, '..', 'tests', 'golden');
let s = 424242;
function rnd() { s = (Math.imul(s, 1103515245) + 12345) >>> 0; return s / 4294967296; }
const pos = () => Math.exp((rnd() - 0.5) * 8), real = () => (rnd() - 0.5) * 20, unit = () => rnd(), cnt = (m) => Math.floor(rnd() * m);
const F = [
  (x, a, b) => ld.norm(x, a, b), (x, a, b) => ld.unif(x, a, b), (x, a, b) => ld.beta(x, a, b), (x, a) => ld.bern(x, a), (x, a) => ld.pois(x, a),
  (x, a, b) => ld.cauchy(x, a, b), (x, a, b) => ld.laplace(x, a, b), (x, a, b) => ld.gamma(x, a, b), (x, a, b) => ld.invgamma(x, a, b),
  (x, a, b) => ld.lnorm(x, a, b), (x, a, b) => ld.pareto(x, a, b), (x, a, b, c) => ld.t(x, a, b, c), (x, a, b) => ld.weibull(x, a, b),
  (x, a, b) => ld.logis(x, a, b), (x, a) => ld.exp(x, a), (x, a, b) => ld.binom(x, a, b), (x, a, b) => ld.nbinom(x, a, b),
  (x, a, b, c) => ld.hyper(x, a, b, c), (x) => ld.lgamma(x), (x) => ld.lfactorial(x), (x, a) => ld.lchoose(x, a), (x, a) => ld.lbeta(x, a)];
const ARGS = [
  () => [real(), real(), pos()], () => [real(), -5, 5 + pos()], () => [rnd() < 0.1 ? real() : unit(), rnd() < 0.1 ? 1 : pos(), rnd() < 0.1 ? 1 : pos()],
  () => [rnd() < 0.1 ? 0.5 : cnt(2), unit()], () => [cnt(60) - 2, pos() * 3],
  () => [real(), real(), pos()], () => [real(), real(), pos()], () => [rnd() < 0.1 ? 0 : (rnd() < 0.1 ? -1 : pos()), rnd() < 0.2 ? 1 : pos(), pos()],
  () => [rnd() < 0.1 ? -pos() : pos(), pos(), pos()], () => [rnd() < 0.1 ? -pos() : pos(), real() / 4, pos()], () => [pos() * 2, pos(), pos()],
  () => [real(), real(), pos(), rnd() < 0.05 ? 1e101 : pos() * 5], () => [rnd() < 0.1 ? 0 : (rnd() < 0.1 ? -1 : pos()), pos() / 2 + 0.1, pos()],
  () => [real(), real(), pos()], () => [rnd() < 0.1 ? -1 : pos(), pos()],
  () => { const n = cnt(50); return [cnt(n + 3) - 1, n, rnd() < 0.1 ? cnt(2) : unit()]; }, () => [cnt(40) - 1, rnd() < 0.5 ? cnt(30) + 1 : pos() * 4, unit()],
  () => { const m = cnt(30) + 5, n = cnt(30) + 5, k = cnt(m + n); return [cnt(k + 2) - 1, m, n, k]; },
  () => [pos() * 10], () => [cnt(100) - 1], () => [cnt(60) + 20, cnt(20)], () => [pos() * 5, pos() * 5]];
const PER = 600, recs = [];
for (let id = 0; id < F.length; id++) for (let r = 0; r < PER; r++) {
  const a = ARGS[id](); while (a.length < 4) a.push(0);
  recs.push([id, a[0], a[1], a[2], a[3], F[id](a[0], a[1], a[2], a[3])]);
}
const buf = Buffer.alloc(recs.length * 48);
recs.forEach((r, i) => r.forEach((v, j) => buf.writeDoubleLE(v, i * 48 + j * 8)));
fs.writeFileSync(path.join(OUT, 'ld_values.bin'), buf);

// ---- Math.pow of this V8 (general exponents: ld.t, ld.weibull)
const N = 60000, pb = Buffer.alloc(N * 24);
const sp = [0, -0, 1, -1, 2, 0.5, -0.5, 3, -3, Infinity, -Infinity, NaN, 1e-310, -1e-310, 1e308, 2.5, -2.5, 1023, 1024, -1074, -1075];
for (let i = 0; i < N; i++) {
  let x, y; const m = i % 6;
  if (i < sp.length * sp.length) { x = sp[i % sp.length]; y = sp[Math.floor(i / sp.length)]; }
  else if (m === 0) { x = rnd() * 10; y = (rnd() - 0.5) * 20; }
  else if (m === 1) { x = Math.exp((rnd() - 0.5) * 100); y = (rnd() - 0.5) * 30; }
  else if (m === 2) { x = 1 + (rnd() - 0.5) * 1e-3; y = (rnd() - 0.5) * 1e6; }
  else if (m === 3) { x = -rnd() * 10; y = Math.round((rnd() - 0.5) * 40); }
  else if (m === 4) { x = 1 + rnd() * 5; y = -(rnd() * 50 + 0.5); }
  else { x = Math.exp((rnd() - 0.5) * 1400); y = (rnd() - 0.5) * 4; }
  pb.writeDoubleLE(x, i * 24); pb.writeDoubleLE(y, i * 24 + 8); pb.writeDoubleLE(Math.pow(x, y), i * 24 + 16);
}
fs.writeFileSync(path.join(OUT, 'v8_pow_pairs.bin'), pb);
// ---- Math.log1p / Math.expm1 of this V8
{
  const M = 60000, lb = Buffer.alloc(M * 24);
  const spx = [0, -0, 1, -1, -0.5, 0.41421356237309503, -0.2928932188134524, 1e-10, -1e-10, 1e-20, Infinity, -Infinity, NaN, -2, 709.78, 710, -40, -38.8, 38.8, 0.5, -0.35, 1.04, -1.04, 56 * Math.LN2, 1e18, Math.pow(2, 52), Math.pow(2, 53)];
  for (let i = 0; i < M; i++) {
    let x; const m = i % 6;
    if (i < spx.length) x = spx[i];
    else if (m === 0) x = (rnd() - 0.5) * 4; else if (m === 1) x = (rnd() - 0.5) * 2e-3; else if (m === 2) x = (rnd() - 0.5) * 1500;
    else if (m === 3) x = Math.exp((rnd() - 0.5) * 80) * (rnd() < 0.5 ? -1 : 1); else if (m === 4) x = (rnd() - 0.3) * 3; else x = -1 + Math.exp(-rnd() * 40);
    lb.writeDoubleLE(x, i * 24); lb.writeDoubleLE(Math.log1p(x), i * 24 + 8); lb.writeDoubleLE(Math.expm1(x), i * 24 + 16);
  }
  fs.writeFileSync(path.join(OUT, 'v8_log1p_expm1_pairs.bin'), lb);
}
// ---- Math.tanh / Math.atan / Math.log10 of this V8: records (x, tanh x, atan x, log10 |x|)
{
  const M = 60000, mb = Buffer.alloc(M * 32);
  const spx = [0, -0, 1, -1, 0.5, -0.5, 22, -22, 23, 1e-9, -1e-9, 1e-300, Infinity, -Infinity, NaN, 0.4375, 0.6875, 1.1875, 2.4375, 1e20, -1e20, 100, 10, 1000, 1e-5];
  for (let i = 0; i < M; i++) {
    let x; const m = i % 5;
    if (i < spx.length) x = spx[i];
    else if (m === 0) x = (rnd() - 0.5) * 6; else if (m === 1) x = (rnd() - 0.5) * 60; else if (m === 2) x = (rnd() - 0.5) * 2e-3;
    else if (m === 3) x = Math.exp((rnd() - 0.5) * 200) * (rnd() < 0.5 ? -1 : 1); else x = (rnd() - 0.5) * 2;
    mb.writeDoubleLE(x, i * 32); mb.writeDoubleLE(Math.tanh(x), i * 32 + 8); mb.writeDoubleLE(Math.atan(x), i * 32 + 16); mb.writeDoubleLE(Math.log10(Math.abs(x)), i * 32 + 24);
  }
  fs.writeFileSync(path.join(OUT, 'v8_math2_pairs.bin'), mb);
}
console.log('ld_values.bin:', recs.length, 'records; v8_pow_pairs.bin:', N, 'pairs');
