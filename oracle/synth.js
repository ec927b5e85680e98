'use strict';
/*
 * synth.js -- TEST INFRASTRUCTURE ONLY.
 * Deterministic synthetic data for the BASELINE.json configs (SURVEY.md §8d),
 * built from Philox uniforms with +,-,* and Math.exp only, so the same recipe
 * gives bit-identical arrays in JS (here), Python (bayes.js_amd/synth.py) and
 * C (oracle/amwg_oracle.c uses the arrays handed to it).
 *   z  = (u1+u2+...+u12) - 6      (Irwin-Hall approximation of N(0,1), exact adds)
 * Data streams use Philox "chain" ids counted down from 2^32-1 so they never
 * collide with sampler chains.
 */
const { stream } = require('./philox.js');
const DATA_CHAIN0 = 4294967295;

function zgen(rand) { return function () { let s = rand(); for (let j = 1; j < 12; j++) s += rand(); return s - 6; }; }

function normal(N, data_seed) {               // cfg1-shape/cfg2: x_i ~ N(3,2)
  const z = zgen(stream(data_seed, DATA_CHAIN0));
  const x = new Array(N);
  for (let i = 0; i < N; i++) x[i] = 3 + 2 * z();
  return { x };
}
function bern(N, data_seed) {                 // cfg3: x_i ~ Bernoulli(0.3)
  const r = stream(data_seed, DATA_CHAIN0);
  const x = new Array(N);
  for (let i = 0; i < N; i++) x[i] = r() < 0.3 ? 1 : 0;
  return { x };
}
function hier(N, G, data_seed) {              // cfg4: theta_g ~ N(5,3), y_i ~ N(theta[g_i], 2), g_i = i mod G
  const zt = zgen(stream(data_seed, DATA_CHAIN0 - 1)), z = zgen(stream(data_seed, DATA_CHAIN0));
  const theta = new Array(G), y = new Array(N), g = new Array(N);
  for (let k = 0; k < G; k++) theta[k] = 5 + 3 * zt();
  for (let i = 0; i < N; i++) { g[i] = i % G; y[i] = theta[g[i]] + 2 * z(); }
  return { y, g, G, theta_true: theta };
}
function glm(N, data_seed) {                  // cfg5: Poisson GLM, 7 real columns + change-point indicator
  const K = 7;
  const beta = [0.5, 0.2, -0.1, 0.05, 0.1, -0.2, 0.15, 0.3];
  const cp = Math.floor(0.4 * N);
  const z = zgen(stream(data_seed, DATA_CHAIN0)), r = stream(data_seed, DATA_CHAIN0 - 1);
  const X = new Array(N * K), y = new Array(N);
  for (let i = 0; i < N; i++) {
    X[i * K] = 1;
    for (let k = 1; k < K; k++) X[i * K + k] = 0.5 * z();
    let eta = 0;
    for (let k = 0; k < K; k++) eta += X[i * K + k] * beta[k];
    if (i >= cp) eta += beta[7];
    const lam = Math.exp(eta), u = r();
    let n = 0, p = Math.exp(-lam), F = p;
    while (u > F && n < 1000) { n++; p = p * lam / n; F += p; }
    y[i] = n;
  }
  return { X, y, K, cp_true: cp, beta_true: beta };
}
module.exports = { normal, bern, hier, glm };
