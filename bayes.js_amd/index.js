'use strict';
// Entry point: `const { mcmc, ld } = require('bayes.js_amd')` mirrors the reference's two
// globals (mcmc.js:9-22 exports `mcmc`, distributions.js:43-56 exports `ld`).
const mcmc = require('./mcmc.js');
module.exports = { mcmc, ld: mcmc.ld, models: mcmc.models };
