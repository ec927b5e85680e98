// The stepper classes of the reference (mcmc.js:1109-1115), on the GPU: a state object shared by the caller and two steppers.
//   node examples/steppers.js          (needs an MI355X; `make -C bayes.js_amd/csrc` first)
// Same constructors as rasmusab/bayes.js: (completed params of the parameter the stepper moves, the shared state, a log_post of NO
// arguments that reads that state).  The translator finds `state` and `y` by name among the globals (script style), or in
// options.constants when the code lives in a module, as here.
'use strict';
const { mcmc, ld } = require('../bayes.js_amd');
global.ld = ld;

const y = [1.5, -0.3, 2.2, 0.9, 3.1, 1.1, 0.4, 2.7];
const state = { mu: 0, sigma: 1, note: 'entries that are not numbers are ignored' };
const posterior = function () {
  var lp = ld.norm(state.mu, 0, 10) + ld.unif(state.sigma, 0, 50);
  for (var i = 0; i < y.length; i++) lp += ld.norm(y[i], state.mu, state.sigma);
  return lp;
};
const free = { constants: { state, y }, seed: 1 };
const stepMu = new mcmc.RealMetropolisStepper({ mu: { lower: -Infinity, upper: Infinity, dim: [1] } }, state, posterior, free);
const stepSigma = new mcmc.RealMetropolisStepper({ sigma: { lower: 0, upper: Infinity, dim: [1] } }, state, posterior, free);

let sumMu = 0, sumSigma = 0;
const n = 4000;
for (let i = 0; i < n + 1000; i++) {
  stepMu.step();            // moves state.mu (reads state.sigma as it is now)
  stepSigma.step();         // moves state.sigma
  if (i >= 1000) { sumMu += state.mu; sumSigma += state.sigma; }
}
console.log('posterior means: mu', (sumMu / n).toFixed(3), ' sigma', (sumSigma / n).toFixed(3), ' (data mean', (y.reduce((a, b) => a + b) / y.length).toFixed(3) + ')');
console.log('stepper info:', JSON.stringify(stepMu.info()));
stepMu.close(); stepSigma.close();
