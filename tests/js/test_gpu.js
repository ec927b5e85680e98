'use strict';
// GPU tests of the JS front-end through the N-API addon: the README programs run unchanged
// and reproduce the seeded reference runs stored in tests/golden/ bit for bit.
const assert = require('assert');
const fs = require('fs');
const path = require('path');
const { mcmc, ld, models } = require('../../bayes.js_amd');
global.ld = ld;

function golden(name) {
  const untag = (k, v) => (v === '__inf' ? Infinity : v === '__-inf' ? -Infinity : v === '__nan' ? NaN : v === '__-0' ? -0 : v);
  return JSON.parse(fs.readFileSync(path.join(__dirname, '..', 'golden', name + '.json'), 'utf8'), untag);
}
const flat = (v) => { const o = []; (function r(x) { Array.isArray(x) ? x.forEach(r) : o.push(x); })(v); return o; };

function checkAgainstGolden(name, makeSampler, names) {
  const g = golden(name);
  for (const rec of g.chains) {
    const s = makeSampler(g, rec);
    const segs = [];
    for (const seg of g.case.schedule) {
      if (seg.op === 'burn') assert.strictEqual(s.burn(seg.n), undefined);
      else if (seg.op === 'stop') s.stop_adaptation();
      else if (seg.op === 'start') s.start_adaptation();
      else { if (seg.thin) s.thin(seg.thin); segs.push(s.sample(seg.n)); }
    }
    segs.forEach((smp, k) => {
      const want = rec.samples[k];
      assert.deepStrictEqual(Object.keys(smp), names);
      assert.strictEqual(smp[names[0]].length, want.kept);
      want.draws.forEach((row, t) => { let got = []; for (const nm of names) got = got.concat(flat(smp[nm][t])); assert.deepStrictEqual(got, row); });
      const sum = new Array(want.sum.length).fill(0);
      for (let t = 0; t < want.kept; t++) { let j = 0; for (const nm of names) for (const v of flat(smp[nm][t])) sum[j++] += v; }
      assert.deepStrictEqual(sum, want.sum);
    });
    let st = []; for (const nm of names) st = st.concat(flat(s.state[nm]));
    assert.deepStrictEqual(st, rec.final_state);
    const inf = s.info();
    let pls = []; for (const nm of names) pls = pls.concat(flat(inf.steppers[nm]).map((o) => o.prop_log_scale));
    assert.deepStrictEqual(pls, rec.prop_log_scale);
    let acc = []; for (const nm of names) acc = acc.concat(flat(inf.steppers[nm]).map((o) => o.accepts));
    assert.deepStrictEqual(acc, rec.accepts);
    assert.strictEqual(s.diagnostics()[0].uniforms[0], rec.uniforms);
    assert.strictEqual(s.log_post(), rec.log_post);          // host closure at the final state == reference's value
    s.close();
  }
}

// ---- 1. README.md:18-43, verbatim -------------------------------------------------------------
{
  // The heights of the last ten American presidents in cm, from Kennedy to Obama
  var data = [183, 192, 182, 183, 177, 185, 188, 188, 182, 185];
  var params = {
    mu: {type: "real"},
    sigma: {type: "real", lower: 0} };
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
  checkAgainstGolden('cfg1_heights', (g, rec) => new mcmc.AmwgSampler(params, log_post, data,
    { seed: g.case.seed, chain_offset: rec.chain, lanes_per_chain: 1 }), ['mu', 'sigma']);
  // options plumbing end to end: global + per-parameter stepper options, stop/start, thin
  checkAgainstGolden('normal_opts', (g, rec) => new mcmc.AmwgSampler(params, log_post, g.data.x,
    Object.assign({ seed: g.case.seed, chain_offset: rec.chain, lanes_per_chain: 1 }, g.case.options)), ['mu', 'sigma']);
  // unseeded construction works and two unseeded samplers differ, like the reference
  var a = new mcmc.AmwgSampler(params, log_post, data), b = new mcmc.AmwgSampler(params, log_post, data);
  a.burn(50); b.burn(50);
  assert.notDeepStrictEqual(a.state, b.state);
  a.close(); b.close();
}

// ---- 2. README.md:149-164 beta-Bernoulli, verbatim ----------------------------------------------
{
  var log_post = function(state, data) {
    // Start by defining a variable to hold the log posterior initialized to 0
    var log_post = 0;
    log_post += ld.beta(state.theta, 2, 2);
    var n = data.x.length;
    for(var i = 0; i < n; i++) {
      log_post += ld.bern(data.x[i], state.theta)
    }
    return log_post;
  }
  checkAgainstGolden('beta_bern_n2000', (g, rec) => new mcmc.AmwgSampler({ theta: { type: 'real', lower: 0, upper: 1 } }, log_post,
    { x: g.data.x }, { seed: g.case.seed, chain_offset: rec.chain, lanes_per_chain: 1 }), ['theta']);
}

// ---- 3. descriptor models: multidimensional draws come back as nested arrays ----------------------
checkAgainstGolden('hier_small', (g, rec) => new mcmc.AmwgSampler(
  { theta: { type: 'real', dim: [g.data.G] }, mu: { type: 'real' }, sigma: { type: 'real', lower: 0, init: 1 } },
  models.hier_normal(), { y: g.data.y, g: g.data.g, G: g.data.G }, { seed: g.case.seed, chain_offset: rec.chain, lanes_per_chain: 1 }),
  ['theta', 'mu', 'sigma']);
checkAgainstGolden('glm_small', (g, rec) => new mcmc.AmwgSampler(
  { beta: { type: 'real', dim: [8], init: 0 }, cp: { type: 'int', lower: 0, upper: g.data.y.length - 1 } },
  models.pois_glm(), { X: g.data.X, y: g.data.y }, { seed: g.case.seed, chain_offset: rec.chain, lanes_per_chain: 1 }), ['beta', 'cp']);

// ---- 4. many chains, thinning, monitor, sharding over "devices" -------------------------------------
{
  const g = golden('normal_n1000');
  const params = { mu: {}, sigma: { lower: 0 } };
  const mk = (extra) => new mcmc.AmwgSampler(params, models.normal(), g.data.x, Object.assign({ seed: 77, chains: 96, lanes_per_chain: 4 }, extra));
  const one = mk({}), two = mk({ devices: [0, 0, 0] });           // three shards on one GPU == one shard
  one.burn(60); two.burn(60);
  one.thin(4); two.thin(4);
  const s1 = one.sample(30), s2 = two.sample(30);
  assert.ok(s1.mu instanceof Float64Array);
  assert.deepStrictEqual(s1.mu.layout, { kept: 8, len: 1, chains: 96, dim: [1] });
  assert.deepStrictEqual(Array.from(s1.mu), Array.from(s2.mu));
  assert.deepStrictEqual(Array.from(s1.sigma), Array.from(s2.sigma));
  assert.deepStrictEqual(Array.from(one.state.mu), Array.from(two.state.mu));
  assert.deepStrictEqual(Array.from(one.info().steppers.sigma.prop_log_scale), Array.from(two.info().steppers.sigma.prop_log_scale));
  // options.gather: the shards' draws are gathered to one device inside the library (amwg_group_gather_draws) and leave it in ONE copy -- same arrays
  {
    const three = mk({ devices: [0, 0, 0], gather: true, gather_root: 1 });
    three.burn(60); three.thin(4);
    const s3 = three.sample(30);
    assert.deepStrictEqual(Array.from(s3.mu), Array.from(s1.mu));
    assert.deepStrictEqual(Array.from(s3.sigma), Array.from(s1.sigma));
    three.close();
  }
  // summaries of the sharded sampler (per-shard reductions + RCCL all-reduce, amwg_group_*) == those of the single shard
  {
    const close = (a, b, tol) => Math.abs(a - b) <= tol * Math.max(1, Math.abs(b));
    const m1 = one.moments(), m2 = two.moments(), c1 = one.convergence(), c2 = two.convergence(), q1 = one.quantiles([0.025, 0.5, 0.975]), q2 = two.quantiles([0.025, 0.5, 0.975]);
    for (const nm of ['mu', 'sigma']) {
      assert.ok(close(m2[nm].mean[0], m1[nm].mean[0], 1e-13) && close(m2[nm].sd[0], m1[nm].sd[0], 1e-11), 'sharded moments ' + nm);
      assert.ok(close(c2[nm].rhat[0], c1[nm].rhat[0], 1e-10) && close(c2[nm].ess[0], c1[nm].ess[0], 1e-9), 'sharded convergence ' + nm);
      assert.deepStrictEqual(q2[nm], q1[nm], 'sharded quantiles ' + nm);          // the same multiset sorted: identical
    }
  }
  // chain 7 of the many-chain run == a single-chain sampler with chain_offset 7 (reference-shaped output)
  const solo = new mcmc.AmwgSampler(params, models.normal(), g.data.x, { seed: 77, chain_offset: 7, lanes_per_chain: 4, thin: 4 });
  solo.burn(60);
  const ss = solo.sample(30);
  assert.ok(Array.isArray(ss.mu) && ss.mu.length === 8);
  for (let t = 0; t < 8; t++) { assert.strictEqual(ss.mu[t], s1.mu[t * 96 + 7]); assert.strictEqual(ss.sigma[t], s1.sigma[t * 96 + 7]); }
  one.monitor(['sigma']);
  assert.deepStrictEqual(Object.keys(one.sample(3)), ['sigma']);
  const m = one.moments();
  assert.ok(Math.abs(m.sigma.mean[0] - 2) < 0.5 && m.sigma.sd[0] > 0);
  assert.ok(typeof one.step().mu[0] === 'number');
  one.close(); two.close(); solo.close();
}
// ---- 5. a recognised family declared in another parameter order falls through to the translator and still reproduces the reference
{
  const um = require('./user_models.js');
  const m = um.build('readme_normal_swapped');
  const g = golden('user_readme_normal_swapped');
  for (const rec of g.chains) {
    const s = new mcmc.AmwgSampler(m.params, m.log_post, m.data, { seed: g.case.seed, chain_offset: rec.chain, lanes_per_chain: 1 });
    assert.strictEqual(s.model, 'translated');
    let smp = null;
    for (const seg of g.case.schedule) { if (seg.op === 'burn') s.burn(seg.n); else smp = s.sample(seg.n); }
    const want = rec.samples[0];
    want.draws.forEach((row, t) => { let got = []; for (const nm of want.keys) got = got.concat(flat(smp[nm][t])); assert.deepStrictEqual(got, row, 'swapped draw ' + t); });
    let st = []; for (const nm of Object.keys(m.params)) st = st.concat(flat(s.state[nm]));
    assert.deepStrictEqual(st, rec.final_state);
    s.close();
  }
}
// ---- 6. options.group_local (hierarchical family): the opt-in group-local evaluation through the JavaScript host takes the reference's
// decisions (accept counts, uniforms consumed of the seeded reference run) and stays within rounding of its draws; other families refuse it
{
  const g = golden('hier_small'), rec = g.chains[0];
  const s = new mcmc.AmwgSampler(
    { theta: { type: 'real', dim: [g.data.G] }, mu: { type: 'real' }, sigma: { type: 'real', lower: 0, init: 1 } },
    models.hier_normal(), { y: g.data.y, g: g.data.g, G: g.data.G }, { seed: g.case.seed, chain_offset: rec.chain, group_local: true });
  assert.strictEqual(s.info().launch[0].lanes_per_chain, 64);
  let smp = null;
  for (const seg of g.case.schedule) { if (seg.op === 'burn') s.burn(seg.n); else smp = s.sample(seg.n); }
  const inf = s.info();
  let acc = []; for (const nm of ['theta', 'mu', 'sigma']) acc = acc.concat(flat(inf.steppers[nm]).map((o) => o.accepts));
  assert.deepStrictEqual(acc, rec.accepts);
  assert.strictEqual(s.diagnostics()[0].uniforms[0], rec.uniforms);
  rec.samples[0].draws.forEach((row, t) => {
    let got = []; for (const nm of ['theta', 'mu', 'sigma']) got = got.concat(flat(smp[nm][t]));
    got.forEach((v, j) => assert.ok(Math.abs(v - row[j]) <= 1e-9 * Math.max(1, Math.abs(row[j])), 'group-local draw ' + t + ',' + j));
  });
  s.close();
  assert.throws(() => new mcmc.AmwgSampler({ mu: {}, sigma: { lower: 0 } }, models.normal(), golden('normal_n1000').data.x, { group_local: true }), /hierarchical/);
}
// ---- 7. the hierarchical family WRITTEN OUT as a closure, on labels that are not i mod G and five groups: recognised from its source
// (models.recognise), so options.group_local applies to a closure a user wrote; the same bits as the descriptor of the same model, and as
// the translated closure (options.translate) takes the same decisions
{
  const G = 5, N = 333, rnd = (function () { let x = 12345; return () => { x = (x * 1103515245 + 12345) % 2147483648; return x / 2147483648; }; })();
  const g = [], y = [];
  for (let i = 0; i < N; i++) { const k = Math.min(G - 1, Math.floor(rnd() * rnd() * G * 1.7)); g.push(k); y.push(3 + k + 2 * (rnd() + rnd() + rnd() - 1.5)); }
  const closure = function (s, d) {
    let lp = 0;
    lp += ld.norm(s.mu, 0, 100);
    lp += ld.unif(s.sigma, 0, 100);
    for (let k = 0; k < d.G; k++) lp += ld.norm(s.theta[k], s.mu, 10);
    for (let i = 0; i < d.y.length; i++) lp += ld.norm(d.y[i], s.theta[d.g[i]], s.sigma);
    return lp;
  };
  const params = () => ({ theta: { type: 'real', dim: [G] }, mu: { type: 'real' }, sigma: { type: 'real', lower: 0, init: 1 } });
  const run = (lp, opts) => {
    const s = new mcmc.AmwgSampler(params(), lp, { y, g, G }, Object.assign({ seed: 77, chains: 3 }, opts));
    s.burn(150);
    const smp = s.sample(40), model = s.model, acc = flat(s.info().steppers.theta).map((o) => o.accepts);
    s.close();
    return { smp, model, acc };
  };
  const a = run(closure, { group_local: true }), b = run(models.hier_normal(), { group_local: true }), c = run(closure, { translate: true });
  assert.strictEqual(a.model, 'hier_normal');
  assert.strictEqual(c.model, 'translated');
  for (const nm of ['theta', 'mu', 'sigma']) assert.deepStrictEqual(Array.from(a.smp[nm]), Array.from(b.smp[nm]));
  assert.deepStrictEqual(a.acc, c.acc);
  for (const nm of ['theta', 'mu', 'sigma']) assert.deepStrictEqual(Array.from(a.smp[nm]), Array.from(c.smp[nm]));      // same decisions => same draws
}
console.log('gpu frontend ok');
