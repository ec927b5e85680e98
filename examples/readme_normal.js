'use strict';
// The README program of rasmusab/bayes.js (README.md:18-43), unchanged except for the require line:
//   node examples/readme_normal.js
const { mcmc, ld } = require('../bayes.js_amd');
global.ld = ld;

// The heights of the last ten American presidents in cm, from Kennedy to Obama
var data = [183, 192, 182, 183, 177, 185, 188, 188, 182, 185];

var params = {
  mu: {type: "real"},
  sigma: {type: "real", lower: 0}};

var log_post = function(state, data) {
  var log_post = 0;
  // Priors
  log_post += ld.norm(state.mu, 0, 100);
  log_post += ld.unif(state.sigma, 0, 100);
  // Likelihood
  for(var i = 0; i < data.length; i++) {
    log_post += ld.norm(data[i], state.mu, state.sigma);
  }
  return log_post;
};

// Initializing the sampler and generate a sample of size 5000 (one chain, like the reference)
var sampler = new mcmc.AmwgSampler(params, log_post, data);
sampler.burn(1000);
var samples = sampler.sample(5000);
const mean = (a) => a.reduce((s, v) => s + v, 0) / a.length;
console.log('1 chain    : mean(mu) = %s  mean(sigma) = %s', mean(samples.mu).toFixed(2), mean(samples.sigma).toFixed(2));
sampler.close();

// The same model on 65 536 chains: summaries are computed on the device, nothing is copied back
var many = new mcmc.AmwgSampler(params, log_post, data, { chains: 65536, seed: 1 });
many.burn(1000);
many.sample_on_device(200);
console.log('65536 chains:', JSON.stringify(many.moments()), JSON.stringify(many.quantiles([0.025, 0.5, 0.975])), JSON.stringify(many.convergence()));
many.close();
