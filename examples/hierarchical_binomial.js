'use strict';
// A closure that is NOT one of the built-in families (tests/test_data.js:174-211 of the reference): it is translated to HIP
// and compiled with hiprtc at construction.  Helper functions of the closure are passed in options.helpers.
//   node examples/hierarchical_binomial.js
const { mcmc, ld } = require('../bayes.js_amd');
global.ld = ld;

var binom_data = {"x": [5, 6, 9, 14, 13, 20], "n": [10, 10, 20, 20, 30, 30]};
var params = {
  "p": {"type": "real", "init": 0.5, "lower": 0, "upper": 1, "dim": [1, 6]},
  "mu_logit_p": {"type": "real", "init": 0},
  "sigma_logit_p": {"type": "real", "lower": 0, "init": 1}};
var logit = function(p) { return Math.log(p / (1 - p)); };
var log_post = function(par, d) {
  var p = par.p[0];
  var mu_logit_p = par.mu_logit_p;
  var sigma_logit_p = par.sigma_logit_p;
  var log_post = 0;
  log_post += ld.norm(mu_logit_p, 0, 10);
  log_post += ld.norm(sigma_logit_p, 0, 10);
  for(var i = 0; i < d.x.length; i++) {
    log_post += ld.norm(logit(p[i]), mu_logit_p, sigma_logit_p);
    log_post += ld.binom(d.x[i], d.n[i], p[i]);
  }
  par.mean_p = (p[0] + p[1] + p[2] + p[3] + p[4] + p[5]) / 6;    // a derived quantity: recorded with the parameters
  return log_post;
};
global.logit = logit;   // only for the host-side evaluation sampler.log_post()

var sampler = new mcmc.AmwgSampler(params, log_post, binom_data, { helpers: { logit: logit }, chains: 4096, seed: 7 });
console.log('model:', sampler.model, '  lanes per chain:', sampler.info().launch[0].lanes_per_chain);
sampler.burn(2000);
sampler.thin(5);
sampler.sample_on_device(1000);
console.log('posterior means :', JSON.stringify(sampler.moments().p.mean.map((v) => +v.toFixed(3))), ' mean_p:', sampler.moments().mean_p.mean[0].toFixed(3));
console.log('raw proportions :', JSON.stringify(binom_data.x.map((x, i) => +(x / binom_data.n[i]).toFixed(3))));
console.log('R-hat           :', JSON.stringify(sampler.convergence().p.rhat.map((v) => +v.toFixed(3))));
sampler.close();
