'use strict';
// Chains sharded over the GPUs of a node, summaries reduced inside the library (RCCL all-reduce), geometry measured at construction:
//   node examples/multi_gpu.js [devices]        e.g.  node examples/multi_gpu.js 0,1,2,3,4,5,6,7     (default "0,0": two shards on GPU 0)
// The model is a random-effects regression written as an ordinary bayes.js closure (it is translated to HIP and compiled at construction).
const { mcmc, ld } = require('../bayes.js_amd');
global.ld = ld;

const devices = (process.argv[2] || '0,0').split(',').map(Number);
let s = 12345;
const rnd = () => { s = (Math.imul(s, 1103515245) + 12345) >>> 0; return s / 4294967296; };
const gauss = () => { let t = 0; for (let j = 0; j < 12; j++) t += rnd(); return t - 6; };
const G = 16, N = 4000, data = { y: [], g: [], x: [] };
const theta = Array.from({ length: G }, () => 2 + 1.5 * gauss());
for (let i = 0; i < N; i++) { const g = i % G, x = rnd() * 2 - 1; data.g.push(g); data.x.push(x); data.y.push(theta[g] + 0.7 * x + 0.5 * gauss()); }

const params = { theta: { dim: [G], init: 0 }, slope: {}, mu: {}, tau: { lower: 0, init: 1 }, sigma: { lower: 0, init: 1 } };
const log_post = function (s, d) {
  let lp = ld.norm(s.mu, 0, 10) + ld.gamma(s.tau, 2, 1) + ld.gamma(s.sigma, 2, 2) + ld.norm(s.slope, 0, 5);
  for (let k = 0; k < s.theta.length; k++) lp += ld.norm(s.theta[k], s.mu, s.tau);
  for (let i = 0; i < d.y.length; i++) lp += ld.norm(d.y[i] - s.slope * d.x[i], s.theta[d.g[i]], s.sigma);
  return lp;
};

const sampler = new mcmc.AmwgSampler(params, log_post, data, { chains: 8192, seed: 7, devices, lanes_per_chain: -2 /* AMWG_LANES_AUTOTUNE */ });
sampler.burn(1500);
sampler.sample_on_device(400);                        // draws stay in HBM, one shard per device
const m = sampler.moments(), c = sampler.convergence(), q = sampler.quantiles([0.025, 0.975]);
const fmt = (v) => v.toFixed(3);
console.log('shards:', sampler.info().launch.map((l) => 'device ' + l.device + ': ' + l.chains + ' chains, ' + l.lanes_per_chain + ' lanes per chain').join(' | '));
for (const name of ['slope', 'mu', 'tau', 'sigma'])
  console.log(name.padEnd(6), 'mean', fmt(m[name].mean[0]), ' 95%', '[' + q[name][0].map(fmt).join(', ') + ']', ' R-hat', fmt(c[name].rhat[0]));
console.log('theta[0..3] mean', m.theta.mean.slice(0, 4).map(fmt).join(' '), ' (generated from', theta.slice(0, 4).map(fmt).join(' ') + ')');
sampler.close();
