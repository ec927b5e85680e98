/*
 * stepper_cases.js -- scenarios for the stand-alone stepper classes (mcmc.js:1109-1115), written once and run twice:
 *   oracle/gen_stepper_golden.js  with the UNMODIFIED reference module, Math.random = a Philox stream per stepper
 *   tests/js/test_gpu_steppers.js with bayes.js_amd (one device sampler per stepper, same Philox streams)
 * The densities and parameter definitions are those of the reference's own stepper tests (tests/test_mcmc_js.R:52-140,
 * 188-220 with tests/test_data.js:93-171), plus shared-state scenarios (several steppers, and the caller, moving one object).
 *
 * ctx = { mcmc, ld, make(Class, params, state, log_post, options, streamId, free) }: `free` names the closure's free
 * variables ({constants, helpers}) -- the reference ignores it, the translator needs it for module-scoped code.
 */
'use strict';
module.exports = function (ctx) {
  const mcmc = ctx.mcmc, ld = ctx.ld;
  const copy = (v) => JSON.parse(JSON.stringify(v));
  const cases = {};

  // the reference's test densities (tests/test_data.js:93-135), kept script-style: they take the state as `par`
  const norm_dens = function (par) { return ld.norm(par.x, 10, 5); };
  const poisson_dens = function (par) { return ld.pois(par.x, 10); };
  const multivar_norm_dens = function (par) {
    var x1 = par.x[0][0];
    var x2 = par.x[0][1];
    var x3 = par.x[1][0];
    var x4 = par.x[1][1];
    var log_post = ld.norm(x1, 1000, 50) + ld.norm(x2, 10, 5) + ld.norm(x3, 0.1, 0.5) + ld.norm(x4, 0.001, 0.05);
    return log_post;
  };
  const multivar_poisson_dens = function (par) {
    var x1 = par.x[0][0];
    var x2 = par.x[0][1];
    var x3 = par.x[1][0];
    var x4 = par.x[1][1];
    var log_post = ld.pois(x1, 0.1) + ld.pois(x2, 10) + ld.pois(x3, 1000) + ld.pois(x4, 100000);
    return log_post;
  };
  const bern_dens = function (par) { return ld.bern(par.x, 0.85); };
  const multi_bern_dens = function (par) {
    var x1 = par.x[0][0];
    var x2 = par.x[0][1];
    var x3 = par.x[1][0];
    var x4 = par.x[1][1];
    return Math.log(x1 * x2 * 0.85 + (1 - x1 * x2) * 0.15) + Math.log(x3 * x4 * 0.75 + (1 - x3 * x4) * 0.25);
  };

  cases.real = function () {      // test_mcmc_js.R:52-65
    const state = { x: 0 };
    const posterior = function () { return norm_dens(state); };
    const st = ctx.make(mcmc.RealMetropolisStepper, { x: { lower: -Infinity, upper: Infinity, dim: [1] } }, state, posterior, undefined, 1,
      { constants: { state }, helpers: { norm_dens } });
    const ret = [];
    for (let i = 0; i < 400; i++) ret.push(st.step());
    return { ret, state: copy(state), info: st.info() };
  };

  cases.int = function () {       // test_mcmc_js.R:67-81
    const state = { x: 1 };
    const posterior = function () { return poisson_dens(state); };
    const st = ctx.make(mcmc.IntMetropolisStepper, { x: { lower: 0, upper: Infinity, dim: [1] } }, state, posterior, undefined, 2,
      { constants: { state }, helpers: { poisson_dens } });
    const ret = [];
    for (let i = 0; i < 400; i++) ret.push(st.step());
    return { ret, state: copy(state), info: st.info() };
  };

  cases.multi_real = function () {   // test_mcmc_js.R:83-100, incl. the stop/start_adaptation sequence
    const state = { x: [[0, 0], [0, 0]] };
    const posterior = function () { return multivar_norm_dens(state); };
    const options = { max_adaptation: 0.2, prop_log_scale: [[10, 0], [-10, 5]] };
    const st = ctx.make(mcmc.MultiRealComponentMetropolisStepper, { x: { lower: -Infinity, upper: Infinity, dim: [2, 2] } }, state, posterior, options, 3,
      { constants: { state }, helpers: { multivar_norm_dens } });
    const ret = [];
    for (let i = 0; i < 100; i++) ret.push(copy(st.step()));
    st.stop_adaptation();
    const info_stopped_before = copy(st.info());
    for (let i = 0; i < 100; i++) ret.push(copy(st.step()));
    const info_stopped_after = copy(st.info());
    st.start_adaptation();
    for (let i = 0; i < 300; i++) ret.push(copy(st.step()));
    return { ret, state: copy(state), info: st.info(), info_stopped_before, info_stopped_after };
  };

  cases.multi_int = function () {    // test_mcmc_js.R:102-121
    const state = { x: [[0, 0], [0, 0]] };
    const posterior = function () { return multivar_poisson_dens(state); };
    const options = { batch_size: 10, target_accept_rate: [[0.22, 0.22], [0.75, 0.10]], prop_log_scale: [[1, 10], [30, 1]] };
    const st = ctx.make(mcmc.MultiIntComponentMetropolisStepper, { x: { lower: 0, upper: Infinity, dim: [2, 2] } }, state, posterior, options, 4,
      { constants: { state }, helpers: { multivar_poisson_dens } });
    const ret = [];
    for (let i = 0; i < 300; i++) ret.push(copy(st.step()));
    return { ret, state: copy(state), info: st.info() };
  };

  cases.binary = function () {       // test_mcmc_js.R:123-130
    const state = { x: 0 };
    const posterior = function () { return bern_dens(state); };
    const st = ctx.make(mcmc.BinaryStepper, { x: { type: 'binary' } }, state, posterior, undefined, 5, { constants: { state }, helpers: { bern_dens } });
    const ret = [];
    for (let i = 0; i < 300; i++) ret.push(st.step());
    return { ret, state: copy(state) };
  };

  cases.binary_component = function () {   // test_mcmc_js.R:132-142
    const state = { x: [[0, 0], [0, 0]] };
    const posterior = function () { return multi_bern_dens(state); };
    const st = ctx.make(mcmc.BinaryComponentStepper, { x: { type: 'binary', dim: [2, 2] } }, state, posterior, undefined, 6,
      { constants: { state }, helpers: { multi_bern_dens } });
    const ret = [];
    for (let i = 0; i < 300; i++) ret.push(copy(st.step()));
    return { ret, state: copy(state) };
  };

  // AmwgStepper on the Normal model (test_mcmc_js.R:188-203): the closure forwards the state AND a data array
  cases.amwg_normal = function () {
    const norm_post = function (par, data) {
      var mu = par.mu;
      var sigma = par.sigma;
      var log_post = 0;
      log_post += ld.norm(mu, 0, 100);
      log_post += ld.unif(sigma, 0, 100);
      for (var i = 0; i < data.length; i++) {
        log_post += ld.norm(data[i], mu, sigma);
      }
      par.var = sigma * sigma;
      return log_post;
    };
    const pars = mcmc.complete_params({ mu: { type: 'real' }, sigma: { type: 'real', lower: 0 } }, mcmc.param_init_fixed);
    const state = { mu: pars.mu.init, sigma: pars.sigma.init };
    const norm_data = [100, 62, 96, 122, 141, 144, 74, 73, 78, 128];
    const posterior = function () { return norm_post(state, norm_data); };
    const st = ctx.make(mcmc.AmwgStepper, pars, state, posterior, undefined, 7, { constants: { state, norm_data }, helpers: { norm_post } });
    const ret = [];
    for (let i = 0; i < 400; i++) { st.step(); ret.push([state.mu, state.sigma, state.var]); }
    return { ret, state: copy(state), info: st.info() };
  };

  // AmwgStepper on the reference's "complex model" (test_mcmc_js.R:205-220, test_data.js:138-171): real + int + binary
  cases.amwg_complex = function () {
    const complex_model_post = function (par, x) {
      var p1 = par.p1;
      var n1 = par.n1;
      var m = par.m;
      var log_post = 0;
      log_post += ld.bern(m, 0.4);
      log_post += ld.beta(p1, 2, 2);
      log_post += ld.nbinom(n1, 2, 0.1);
      for (var i = 0; i < x.length; i++) {
        if (m === 0) {
          log_post += ld.nbinom(x[i], 21, 0.5);
        } else {
          log_post += ld.nbinom(x[i], n1, p1);
        }
      }
      return log_post;
    };
    const pars = mcmc.complete_params({ p1: { type: 'real', lower: 0, upper: 1 }, n1: { type: 'int', lower: 1, init: 1 }, m: { type: 'binary' } }, mcmc.param_init_fixed);
    const state = { m: pars.m.init, p1: pars.p1.init, n1: pars.n1.init };
    const nbinom_data = [9, 8, 32, 14, 10, 18, 15, 16, 15, 19];
    const posterior = function () { return complex_model_post(state, nbinom_data); };
    const st = ctx.make(mcmc.AmwgStepper, pars, state, posterior, undefined, 8, { constants: { state, nbinom_data }, helpers: { complex_model_post } });
    const ret = [];
    for (let i = 0; i < 400; i++) { st.step(); ret.push([state.m, state.n1, state.p1]); }
    return { ret, state: copy(state) };
  };

  // one state object, three movers: a Real stepper on mu, a Real stepper on sigma (lower 0), and the caller, who rewrites the entry
  // `shift` and an element of the array `w` between steps.  The closure reads the state directly (no forwarding).
  cases.shared_state = function () {
    const state = { mu: 1, sigma: 2, shift: 0.5, w: [1, 2, 0.5], label: 'not a number' };
    const y = [1.5, -0.3, 2.2, 0.9, 3.1, 1.1];
    const posterior = function () {
      var lp = ld.norm(state.mu, 0, 10) + ld.unif(state.sigma, 0, 50);
      for (var i = 0; i < y.length; i++) lp += state.w[i % 3] * ld.norm(y[i] - state.shift, state.mu, state.sigma);
      return lp;
    };
    const free = { constants: { state, y } };
    const a = ctx.make(mcmc.RealMetropolisStepper, { mu: { lower: -Infinity, upper: Infinity, dim: [1] } }, state, posterior, { batch_size: 7 }, 9, free);
    const b = ctx.make(mcmc.RealMetropolisStepper, { sigma: { lower: 0, upper: Infinity, dim: [1] } }, state, posterior, { prop_log_scale: -1 }, 10, free);
    const ret = [];
    for (let i = 0; i < 150; i++) {
      const r1 = a.step(), r2 = b.step();
      if (i % 10 === 3) state.shift = 0.5 + i / 100;
      if (i % 25 === 7) state.w[1] = 2 + i / 50;
      if (i === 60) state.mu = -4;            // the caller moves a stepped entry as well
      ret.push([r1, r2, state.mu, state.sigma]);
    }
    return { ret, state: copy(state), info_a: a.info(), info_b: b.info() };
  };

  return cases;
};
