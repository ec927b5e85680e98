// A model written the way one writes JavaScript today -- rows of records, a categorical column, helper functions that take the state,
// destructuring, for-of, reduce -- translated to HIP and compiled at construction, then 4096 chains on the GPU.
//   node examples/records_model.js          (needs an MI355X; `make -C bayes.js_amd/csrc` first)
'use strict';
const { mcmc, ld } = require('../bayes.js_amd');
global.ld = ld;

// a small two-arm trial with a covariate, rows as they come out of a CSV parser
let seed = 7;
const rnd = () => { seed = (Math.imul(seed, 1103515245) + 12345) >>> 0; return seed / 4294967296; };
const gauss = () => Math.sqrt(-2 * Math.log(rnd() + 1e-12)) * Math.cos(2 * Math.PI * rnd());
const rows = [];
for (let i = 0; i < 200; i++) {
  const arm = i % 2 ? 'treated' : 'control', age = 30 + 40 * rnd();
  rows.push({ arm, age, y: 1.0 + (arm === 'treated' ? 0.8 : 0) + 0.02 * (age - 50) + 0.7 * gauss() });
}

const log_prior = ({ base, effect, slope, sigma }) => ld.norm(base, 0, 10) + ld.norm(effect, 0, 5) + ld.norm(slope, 0, 1) + ld.cauchy(sigma, 0, 2);
const expected = (s, row) => s.base + (row.arm === 'treated' ? s.effect : 0) + s.slope * (row.age - 50);
const log_post = (s, d) => {
  let lp = log_prior(s);
  for (const row of d.rows) lp += ld.norm(row.y, expected(s, row), s.sigma);
  s.relative_effect = s.effect / s.base;        // a derived quantity: recorded with the draws
  return lp;
};

const params = { base: {}, effect: {}, slope: {}, sigma: { lower: 0, init: 1 } };
const sampler = new mcmc.AmwgSampler(params, log_post, { rows }, { chains: 4096, seed: 1, helpers: { log_prior, expected } });
sampler.burn(500);
sampler.thin(5);
sampler.sample_on_device(500);                   // draws stay in HBM; summaries are reduced on the device
const m = sampler.moments(), q = sampler.quantiles([0.025, 0.975]), c = sampler.convergence();
for (const k of ['base', 'effect', 'slope', 'sigma', 'relative_effect'])
  console.log(k.padEnd(16), 'mean', m[k].mean[0].toFixed(3), ' 95% [' + q[k][0].map((v) => v.toFixed(3)).join(', ') + ']', ' R-hat', c[k].rhat[0].toFixed(3));
console.log('geometry:', JSON.stringify(sampler.info().launch[0]));
sampler.close();
