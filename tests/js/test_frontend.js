'use strict';
// Host-logic tests of the JS front-end (no GPU): parameter completion KATs, option merging,
// model recognition, error behaviour.  Expected values are the reference's: the completion
// KATs restate tests/test_data.js:9-74 (checked by tests/test_mcmc_js.R:39-46); option
// merging is compared with what the reference's steppers held in tests/golden/normal_opts.json;
// when /root/reference is present everything is ALSO compared with the live reference.
const assert = require('assert');
const fs = require('fs');
const path = require('path');
const { mcmc, ld, models } = require('../../bayes.js_amd');

const REF = require('../../oracle/ref_dir.js').refDir() || '/root/reference';
const haveRef = fs.existsSync(path.join(REF, 'mcmc.js'));
const ref = haveRef ? require(path.join(REF, 'mcmc.js')) : null;
const refld = haveRef ? require(path.join(REF, 'distributions.js')) : null;

// ---- complete_params KATs
const params1 = { mu: { type: 'real' }, sigma: { type: 'real', lower: 0, init: 1 } };
const params1_completed = { mu: { type: 'real', dim: [1], upper: Infinity, lower: -Infinity, init: 0.5 },
  sigma: { type: 'real', dim: [1], upper: Infinity, lower: 0, init: 1 } };
const params2 = { theta: { init: function () { return 1.5; } }, state: { type: 'binary', init: 1 },
  mat: { type: 'int', dim: [3, 3], init: function () { return 2; } } };
const params2_completed = { theta: { type: 'real', dim: [1], upper: Infinity, lower: -Infinity, init: 1.5 },
  state: { type: 'binary', init: 1, dim: [1], upper: 1, lower: 0 },
  mat: { type: 'int', dim: [3, 3], upper: Infinity, lower: -Infinity, init: [[2, 2, 2], [2, 2, 2], [2, 2, 2]] } };
assert.deepStrictEqual(mcmc.complete_params(params1, mcmc.param_init_fixed), params1_completed);
assert.deepStrictEqual(mcmc.complete_params(params2, mcmc.param_init_fixed), params2_completed);
assert.strictEqual(params1.mu.dim, undefined, 'input must not be modified');
const more = { a: { lower: 2 }, b: { upper: -1 }, c: { lower: 0, upper: 5 }, d: { type: 'int' }, e: { type: 'int', lower: 3 },
  f: { type: 'int', upper: 9 }, g: { type: 'int', lower: 0, upper: 9 }, h: { dim: 4, lower: 1 }, i: { dim: [2, 2], init: 7 } };
const moreDone = mcmc.complete_params(more, mcmc.param_init_fixed);
assert.deepStrictEqual([moreDone.a.init, moreDone.b.init, moreDone.c.init, moreDone.d.init, moreDone.e.init, moreDone.f.init, moreDone.g.init],
  [2.5, -1.5, 2.5, 1, 4, 8, 5]);
assert.deepStrictEqual(moreDone.h.init, [1.5, 1.5, 1.5, 1.5]);
assert.deepStrictEqual(moreDone.h.dim, [4]);
assert.deepStrictEqual(moreDone.i.init, [[7, 7], [7, 7]]);
assert.throws(() => mcmc.param_init_fixed('real', 3, 1), (e) => e === 'Can not initialize parameter where lower bound > upper bound');
if (haveRef) {
  for (const p of [params1, params2, more]) assert.deepStrictEqual(mcmc.complete_params(p, mcmc.param_init_fixed), ref.complete_params(p, ref.param_init_fixed));
}

// ---- option merging (mcmc.js:869-878 `||` semantics + defaults of :500-505)
const gold = JSON.parse(fs.readFileSync(path.join(__dirname, '..', 'golden', 'normal_opts.json'), 'utf8'));
const done = mcmc.complete_params({ mu: { type: 'real' }, sigma: { type: 'real', lower: 0 } }, mcmc.param_init_fixed);
const merged = [].concat(mcmc.componentOptions('mu', done.mu, gold.case.options), mcmc.componentOptions('sigma', done.sigma, gold.case.options));
assert.deepStrictEqual(merged, gold.chains[0].comp_opts);
// falsy per-parameter overrides are ignored (0 || global, false || global); a falsy GLOBAL value survives
const quirk = mcmc.componentOptions('mu', done.mu, { prop_log_scale: 2, is_adapting: true, params: { mu: { prop_log_scale: 0, is_adapting: false } } });
assert.strictEqual(quirk[0].prop_log_scale, 2);
assert.strictEqual(quirk[0].is_adapting, true);
assert.strictEqual(mcmc.componentOptions('mu', done.mu, { is_adapting: false })[0].is_adapting, false);
// multidimensional: scalar broadcast, array of the right shape, wrong shape throws the reference's message
const md = mcmc.complete_params({ b: { dim: [2, 2] } }, mcmc.param_init_fixed).b;
assert.deepStrictEqual(mcmc.componentOptions('b', md, { batch_size: 10 }).map((o) => o.batch_size), [10, 10, 10, 10]);
assert.deepStrictEqual(mcmc.componentOptions('b', md, { params: { b: { target_accept_rate: [[0.1, 0.2], [0.3, 0.4]] } } }).map((o) => o.target_accept_rate), [0.1, 0.2, 0.3, 0.4]);
assert.throws(() => mcmc.componentOptions('b', md, { params: { b: { batch_size: [1, 2, 3] } } }),
  (e) => e === 'The option batch_size is of dimension [3] but should be [2,2].');
if (haveRef) {   // same merge as the live reference's steppers
  const opts = { batch_size: 10, max_adaptation: 0.5, params: { x: { max_adaptation: 0.1, prop_log_scale: [[1, 2], [3, 4]] } } };
  const st = { x: [[0, 0], [0, 0]], y: 1 };
  const prm = ref.complete_params({ x: { dim: [2, 2] }, y: {} }, ref.param_init_fixed);
  const stp = new ref.AmwgStepper(prm, st, () => 0, JSON.parse(JSON.stringify(opts)));
  const flat = []; stp.substeppers[0].substeppers.forEach((r) => r.forEach((s) => flat.push(s))); flat.push(stp.substeppers[1]);
  const want = flat.map((s) => ({ prop_log_scale: s.prop_log_scale, batch_size: s.batch_size, max_adaptation: s.max_adaptation,
    initial_adaptation: s.initial_adaptation, target_accept_rate: s.target_accept_rate, is_adapting: s.is_adapting }));
  const mine = mcmc.complete_params({ x: { dim: [2, 2] }, y: {} }, mcmc.param_init_fixed);
  assert.deepStrictEqual([].concat(mcmc.componentOptions('x', mine.x, opts), mcmc.componentOptions('y', mine.y, opts)), want);
}

// ---- host ld.* equals the reference's outputs (SURVEY.md Appendix B7)
assert.strictEqual(ld.norm(183, 184.4, 4.9), -2.5489900648518664);
assert.strictEqual(ld.unif(5, 0, 100), -4.605170185988091);
assert.strictEqual(ld.pois(3, 10), -4.884004190245918);
assert.strictEqual(ld.beta(0.3, 2, 2), 0.23111172096338728);
assert.strictEqual(ld.bern(1, 0.3), -1.2039728043259361);
if (haveRef) for (const x of [0.1, 0.5, 3, 7.25]) {
  assert.strictEqual(ld.norm(x, 1, 2), refld.norm(x, 1, 2)); assert.strictEqual(ld.pois(Math.floor(x), 2.5), refld.pois(Math.floor(x), 2.5));
  assert.strictEqual(ld.beta(x / 8, 2, 3), refld.beta(x / 8, 2, 3)); assert.strictEqual(ld.lgamma(x), refld.lgamma(x));
}

// ---- model recognition: the README closures verbatim (README.md:26-36, 150-163)
global.ld = ld;
const readme_normal = function(state, data) {
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
const readme_bern = function(state, data) {
  var log_post = 0;
  log_post += ld.beta(state.theta, 2, 2);
  var n = data.x.length;
  for(var i = 0; i < n; i++) {
    log_post += ld.bern(data.x[i], state.theta)
  }
  return log_post;
}
let r = models.recognise(readme_normal);
assert.deepStrictEqual([r.family, r.hyper, r.paramNames], ['normal', [0, 100, 0, 100], ['mu', 'sigma']]);
assert.deepStrictEqual(r.extract([1, 2, 3]), { x: [1, 2, 3] });
r = models.recognise(readme_bern);
assert.deepStrictEqual([r.family, r.hyper, r.paramNames], ['beta_bern', [2, 2], ['theta']]);
r = models.recognise((s, d) => { let lp = 0; lp += ld.norm(s.m, -2.5, 7); lp += ld.unif(s.s, 0.5, 30); for (let i = 0; i < d.obs.length; i++) lp += ld.norm(d.obs[i], s.m, s.s); return lp; });
assert.deepStrictEqual([r.family, r.hyper, r.paramNames], ['normal', [-2.5, 7, 0.5, 30], ['m', 's']]);
for (const bad of [function (s, d) { return Math.sin(s.x); },
  function (s, d) { var lp = 0; for (var i = 0; i < d.length; i++) { lp += ld.norm(d[i], s.mu, s.sigma); } lp += ld.norm(s.mu, 0, 100); lp += ld.unif(s.sigma, 0, 100); return lp; },  // priors after the loop: different summation order
  function (s, d) { var lp = 0; lp += ld.norm(s.mu, 0, 100); lp += ld.unif(s.sigma, 0, 100); for (var i = 0; i < d.length; i++) { lp += ld.norm(d[i], s.mu, 2 * s.sigma); } return lp; }])
  assert.strictEqual(models.recognise(bad), null);
// the hierarchical family written out as a closure (SURVEY.md section 8(d) cfg4) is recognised from its source -- with it `options.group_local`
// is available to a closure a user wrote, not only to mcmc.models.hier_normal() -- in its three spellings of the loop over the group means
const hier_src = (bound) => new Function('ld', 'return function (s, d) { let lp = 0; lp += ld.norm(s.mu, 1, 50); lp += ld.unif(s.sigma, 0, 20); ' +
  'for (let k = 0; k < ' + bound + '; k++) lp += ld.norm(s.theta[k], s.mu, 7); for (let i = 0; i < d.y.length; i++) lp += ld.norm(d.y[i], s.theta[d.g[i]], s.sigma); return lp; }')(ld);
const hdata = { y: [1, 2, 3, 4, 5, 6], g: [0, 1, 2, 0, 1, 2], G: 3 };
for (const bound of ['d.G', '3', 's.theta.length']) {
  r = models.recognise(hier_src(bound));
  assert.deepStrictEqual([r.family, r.hyper, r.paramNames], ['hier_normal', [1, 50, 0, 20, 7], ['theta', 'mu', 'sigma']]);
  assert.deepStrictEqual(r.extract(hdata, { theta: { dim: [3] } }), { x: hdata.y, g: hdata.g, G: 3 });
}
for (const bad of [
  function (s, d) { let lp = 0; for (let k = 0; k < d.G; k++) lp += ld.norm(s.theta[k], s.mu, 7); lp += ld.norm(s.mu, 1, 50); lp += ld.unif(s.sigma, 0, 20); for (let i = 0; i < d.y.length; i++) lp += ld.norm(d.y[i], s.theta[d.g[i]], s.sigma); return lp; },   // another order of the terms
  function (s, d) { let lp = 0; lp += ld.norm(s.mu, 1, 50); lp += ld.unif(s.sigma, 0, 20); for (let k = 0; k < d.G; k++) lp += ld.norm(s.theta[k], s.mu, s.sigma); for (let i = 0; i < d.y.length; i++) lp += ld.norm(d.y[i], s.theta[d.g[i]], s.sigma); return lp; },   // tau is a parameter
  function (s, d) { let lp = 0; lp += ld.norm(s.mu, 1, 50); lp += ld.unif(s.sigma, 0, 20); for (let k = 0; k < d.G; k++) lp += ld.norm(s.theta[k], s.mu, 7); for (let i = 0; i < d.y.length; i++) lp += ld.norm(d.y[i], s.eta[d.g[i]], s.sigma); return lp; }])      // the means are another parameter
  assert.strictEqual(models.recognise(bad), null);
// descriptor closures evaluate like the README closure on the host
const data10 = [183, 192, 182, 183, 177, 185, 188, 188, 182, 185];
assert.strictEqual(models.normal()({ mu: 180, sigma: 5 }, data10), readme_normal({ mu: 180, sigma: 5 }, data10));

// ---- errors (thrown as strings, like the reference) before any device is touched
const params = { mu: { type: 'real' }, sigma: { type: 'real', lower: 0 } };
// a closure outside the translatable subset is refused with a string that says why (no CPU fallback)
assert.throws(() => new mcmc.AmwgSampler(params, (s, d) => Math.random() * s.mu, data10), (e) => typeof e === 'string' && /cannot translate log_post: Math\.random is not supported/.test(e));
assert.throws(() => new mcmc.AmwgSampler(params, function (s, d) { return helper(s.mu); }, data10), (e) => typeof e === 'string' && /'helper' is not defined inside log_post/.test(e));
assert.throws(() => new mcmc.AmwgSampler(params, function (s, d) { s.mu = 1; return 0; }, data10), (e) => typeof e === 'string' && /assigns to the parameter state\.mu/.test(e));
assert.throws(() => new mcmc.AmwgSampler(params, function (s, d) { return s.tau; }, data10), (e) => typeof e === 'string' && /state\.tau is read but it is neither a parameter nor a derived quantity/.test(e));
assert.throws(() => new mcmc.AmwgSampler(params, function (s, d) { for (var k in d) { } return 0; }, data10), (e) => typeof e === 'string' && /for-in/.test(e));
// a recognised family declared in another order than the hand-written kernel's is not refused (the reference takes any order,
// mcmc.js:839): it goes through the translator and gets as far as opening the device, which this CPU-only test does not have
assert.throws(() => new mcmc.AmwgSampler({ sigma: { lower: 0 }, mu: {} }, readme_normal, data10), (e) => e instanceof Error && e.code === 'AMWG_EHIP');
// the Philox key must be an unsigned integer: anything else would silently become another key
for (const bad of [-1, 1.5, NaN, Infinity, 2 ** 53, -5n, 2n ** 64n])
  assert.throws(() => new mcmc.AmwgSampler(params, readme_normal, data10, { seed: bad }), (e) => typeof e === 'string' && /options\.seed must be a non-negative integer/.test(e));
assert.throws(() => new mcmc.AmwgSampler({ mu: { type: 'binary' }, sigma: {} }, readme_normal, data10), (e) => typeof e === 'string' && /has no binary parameters/.test(e));
assert.throws(() => new mcmc.AmwgSampler({ mu: { type: 'complex', init: 1 }, sigma: {} }, readme_normal, data10), (e) => e === "AmwgStepper can't handle parameter mu with type complex");

// ---- the translator: every closure of tests/js/user_models.js becomes HIP text that hiprtc compiles for gfx950
// together with the step kernel (no device needed); values are checked in tests/test_translate.py
{
  const um = require('./user_models.js');
  for (const name of um.names) {
    const m = um.build(name);
    const tr = mcmc.translate(m.log_post, mcmc.complete_params(m.params, mcmc.param_init_fixed), m.data, { helpers: m.helpers, constants: m.constants });
    assert.ok(tr.source.indexOf('struct UserModel') > 0, name);
    assert.ok(mcmc.native().compileUser(tr.source, tr.parallel ? 4 : 1, 256, 'gfx950') > 10000, name);
  }
  // parameter completion and option merging on randomly drawn configurations: what this front-end hands to the C ABI equals what the
  // reference's constructor built (recorded from the live reference objects in tests/golden/user_cfgfuzz_*.json)
  {
    const untag = (k, v) => (v === '__inf' ? Infinity : v === '__-inf' ? -Infinity : v === '__nan' ? NaN : v === '__-0' ? -0 : v);
    const flat = (v) => { const o = []; (function r(x) { Array.isArray(x) ? x.forEach(r) : o.push(x); })(v); return o; };
    let checked = 0;
    for (const name of um.names.filter((n) => /^cfg(fuzz|edge)_/.test(n))) {
      const m = um.build(name), rec = JSON.parse(fs.readFileSync(path.join(__dirname, '..', 'golden', 'user_' + name + '.json'), 'utf8'), untag).chains[0];
      const done = mcmc.complete_params(m.params, mcmc.param_init_fixed);
      assert.deepStrictEqual(Object.keys(done).map((k) => ({ name: k, type: done[k].type, dim: done[k].dim, lower: done[k].lower, upper: done[k].upper, init: flat(done[k].init) })), rec.params_completed, name);
      let ci = 0;
      for (const k of Object.keys(done)) for (const o of mcmc.componentOptions(k, done[k], m.options)) {
        const want = rec.comp_opts[ci++];
        if (done[k].type === 'binary') continue;       // BinarySteppers hold no options (mcmc.js:745-752)
        assert.deepStrictEqual(o, want, name + ' ' + k);
        checked++;
      }
      assert.strictEqual(ci, rec.comp_opts.length);
    }
    assert.ok(checked >= 40);
  }
  // script-style globals (the reference's README / test pages define helpers and constants as globals) are visible to the translator
  {
    global.logit_g = function(p) { return Math.log(p / (1 - p)); };
    global.HYPER_G = [0, 10];
    const lp = function(par, d) { var s = ld.norm(par.m, HYPER_G[0], HYPER_G[1] * 2); for (var i = 0; i < d.x.length; i++) { s += ld.norm(logit_g(d.x[i]), par.m, 1); } return s; };
    const trg = mcmc.translate(lp, mcmc.complete_params({ m: {} }, mcmc.param_init_fixed), { x: [0.2, 0.5, 0.7] }, {});
    // (round 6: a density with literal parameters has its constants folded at translation time -- ld_norm_c(x, mean, c, den) with the call it stands for in a comment)
    assert.ok(trg.source.indexOf('h_logit_g') > 0 && /ld_norm_c\(S\(0\), 0\.0, [^)]*\) \/\* ld_norm\(\., \., 20\.0\) \*\//.test(trg.source));
    delete global.logit_g; delete global.HYPER_G;
  }
  // README closures translate to lane-split, LDS-staged, hoisted code
  const tr = mcmc.translate(readme_normal, mcmc.complete_params(params, mcmc.param_init_fixed), data10, {});
  assert.strictEqual(tr.parallel, 1);
  assert.strictEqual(tr.lds_bytes, 16);            // ten integer heights: stored as u8
  // the canonical likelihood loop compiles to the hand-scheduled pass (csrc/amwg_pass.h via norm_data_loop) with the hoisted sd-invariants
  assert.ok(/norm_inv\(S\(1\)\)/.test(tr.source) && /norm_data_loop<G>\(A0, static_cast<const uint8_t \*>\(user_arr<0>\(d\)\), 10, S\(0\), k0, true, sub, v_log_post\)/.test(tr.source));
  // ... unless asked not to: then the generic lane-split loop with its fast / IEEE pair
  const trg2 = mcmc.translate(readme_normal, mcmc.complete_params(params, mcmc.param_init_fixed), data10, { no_staged_norm: true });
  assert.ok(/ld_norm_fast\(\(double\)A0\[v_i\], S\(0\), k0, rlo_, rhi_\)/.test(trg2.source) && /ld_norm_slow/.test(trg2.source));
  // every host-side ld.* equals the reference's value on the committed argument sets (tests/golden/ld_values.bin)
  const b = fs.readFileSync(path.join(__dirname, '..', 'golden', 'ld_values.bin'));
  const fnames = ['norm', 'unif', 'beta', 'bern', 'pois', 'cauchy', 'laplace', 'gamma', 'invgamma', 'lnorm', 'pareto', 't', 'weibull', 'logis', 'exp', 'binom', 'nbinom', 'hyper', 'lgamma', 'lfactorial', 'lchoose', 'lbeta'];
  let n = 0;
  for (let i = 0; i < b.length; i += 48) {
    const r = [0, 1, 2, 3, 4, 5].map((j) => b.readDoubleLE(i + j * 8));
    assert.ok(Object.is(ld[fnames[r[0]]](r[1], r[2], r[3], r[4]), r[5]), fnames[r[0]]);
    n++;
  }
  assert.strictEqual(n, 13200);
  if (haveRef) assert.deepStrictEqual(Object.keys(refld).filter((k) => !(k in ld)), []);
}
// the addon loads and, without a GPU, construction fails loudly (no JS fallback)
const nat = mcmc.native();
assert.ok(/gfx950/.test(nat.version()));
assert.strictEqual(nat.mathExp(1), Math.exp(1));
assert.strictEqual(nat.mathLog(0.3), Math.log(0.3));
if (!fs.existsSync('/dev/kfd')) assert.throws(() => new mcmc.AmwgSampler(params, readme_normal, data10, { seed: 1 }), (e) => /no HIP device/.test(e.message));
// host-side helpers exported like the reference's (mcmc.js:1104-1106): same values under the same Math.random stream
if (haveRef) {
  const { stream } = require('../../oracle/philox.js');
  const saved = Math.random;
  try {
    Math.random = stream(5, 0); const a = [mcmc.runif(2, 5), mcmc.runif_discrete(1, 6), mcmc.rnorm(3, 2), mcmc.rnorm(0, 1)];
    Math.random = stream(5, 0); const b = [ref.runif(2, 5), ref.runif_discrete(1, 6), ref.rnorm(3, 2), ref.rnorm(0, 1)];
    assert.deepStrictEqual(a, b);
  } finally { Math.random = saved; }
}
console.log('frontend ok' + (haveRef ? ' (also checked against the live reference)' : ''));
