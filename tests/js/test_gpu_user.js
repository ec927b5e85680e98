'use strict';
// GPU tests of TRANSLATED closures (bayes.js_amd/translate.js + hiprtc): every closure of user_models.js runs through
// mcmc.AmwgSampler with one lane per chain and must reproduce the seeded run of the UNMODIFIED reference stored in
// tests/golden/user_*.json bit for bit -- draws (derived quantities included), final state, adaptation state,
// accept / flip counts, uniforms consumed -- for real, int and binary parameters.
const assert = require('assert');
const fs = require('fs');
const path = require('path');
const { mcmc, ld } = require('../../bayes.js_amd');
const um = require('./user_models.js');
global.ld = ld;

function golden(name) {
  const untag = (k, v) => (v === '__inf' ? Infinity : v === '__-inf' ? -Infinity : v === '__nan' ? NaN : v === '__-0' ? -0 : v);
  return JSON.parse(fs.readFileSync(path.join(__dirname, '..', 'golden', name + '.json'), 'utf8'), untag);
}
const flat = (v) => { const o = []; (function r(x) { Array.isArray(x) ? x.forEach(r) : o.push(x); })(v); return o; };
const only = process.argv.slice(2);

for (const name of um.names) {
  if (only.length && only.indexOf(name) < 0) continue;
  const m = um.build(name);
  for (const k of Object.keys(m.helpers || {})) global[k] = m.helpers[k];      // the host-side closure needs them too
  for (const k of Object.keys(m.constants || {})) global[k] = m.constants[k];
  const g = golden('user_' + name);
  const names = Object.keys(m.params);
  for (const rec of g.chains) {
    const s = new mcmc.AmwgSampler(m.params, m.log_post, m.data, Object.assign({}, m.options,      // m.options: the fixture's stepper options (global / per parameter / per component)
      { seed: g.case.seed, chain_offset: rec.chain, lanes_per_chain: 1, translate: true, helpers: m.helpers, constants: m.constants }));
    assert.strictEqual(s.model, 'translated');
    const segs = [];
    for (const seg of g.case.schedule) {
      if (seg.op === 'burn') s.burn(seg.n);
      else if (seg.op === 'stop') s.stop_adaptation();
      else if (seg.op === 'start') s.start_adaptation();
      else { if (seg.thin) s.thin(seg.thin); segs.push(s.sample(seg.n)); }
    }
    segs.forEach((smp, k) => {
      const want = rec.samples[k];
      assert.deepStrictEqual(Object.keys(smp), want.keys, name);        // parameters, then derived quantities
      assert.strictEqual(smp[names[0]].length, want.kept);
      want.draws.forEach((row, t) => { let got = []; for (const nm of want.keys) got = got.concat(flat(smp[nm][t])); assert.deepStrictEqual(got, row, name + ' draw ' + t); });
      const sum = new Array(want.sum.length).fill(0);
      for (let t = 0; t < want.kept; t++) { let j = 0; for (const nm of want.keys) for (const v of flat(smp[nm][t])) sum[j++] += v; }
      assert.deepStrictEqual(sum, want.sum, name);
    });
    const st = s.state;
    let stv = []; for (const nm of names) stv = stv.concat(flat(st[nm]));
    assert.deepStrictEqual(stv, rec.final_state, name);
    const inf = s.info();
    const per = (key) => { let o = []; for (const nm of names) o = o.concat(flat(inf.steppers[nm]).map((x) => (x[key] === undefined ? 0 : x[key]))); return o; };   // binary: no such field, as in the harness
    assert.deepStrictEqual(per('accepts'), rec.accepts, name + ' accepts');
    assert.deepStrictEqual(per('inbounds'), rec.inbounds, name + ' inbounds');
    assert.deepStrictEqual(per('prop_log_scale'), rec.prop_log_scale, name);
    assert.deepStrictEqual(per('batch_count'), rec.batch_count, name);
    const dg = s.diagnostics()[0];
    assert.strictEqual(dg.uniforms[0], rec.uniforms, name);
    assert.ok(Object.is(dg.log_post[0], rec.log_post), name + ' cached log_post');
    assert.ok(Object.is(s.log_post(), rec.log_post), name + ' host log_post');
    s.close();
  }
  console.log('ok', name);
}

// ---- many chains, G lanes per chain: chain c of a many-chain run == the same chain run alone; moments are sane;
// a closure that cannot be split refuses lanes_per_chain > 1
if (!only.length) {
  const m = um.build('norm_post_derived');
  const many = new mcmc.AmwgSampler(m.params, m.log_post, m.data, { seed: 5, chains: 200, lanes_per_chain: 4, translate: true });
  many.burn(300);
  const smp = many.sample(100);
  assert.deepStrictEqual(smp.var.layout, { kept: 100, len: 1, chains: 200, dim: [1] });
  for (let t = 0; t < 100; t += 9) for (let c = 0; c < 200; c += 37) assert.strictEqual(smp.var[t * 200 + c], smp.sigma[t * 200 + c] * smp.sigma[t * 200 + c]);
  const mom = many.moments();
  assert.ok(Math.abs(mom.mu.mean[0] - 101.8) < 8 && Math.abs(mom.sigma.mean[0] - 37) < 8 && mom.var.mean[0] > 500, JSON.stringify(mom));
  const solo = new mcmc.AmwgSampler(m.params, m.log_post, m.data, { seed: 5, chain_offset: 123, lanes_per_chain: 4, translate: true });
  solo.burn(300);
  const one = solo.sample(100);
  for (let t = 0; t < 100; t++) { assert.strictEqual(one.mu[t], smp.mu[t * 200 + 123]); assert.strictEqual(one.var[t], smp.var[t * 200 + 123]); }
  many.close(); solo.close();
  // three shards on one GPU == one shard (chains are keyed by their global id), binary + int + real parameters
  const cm = um.build('complex_model');
  const mk = (extra) => new mcmc.AmwgSampler(cm.params, cm.log_post, cm.data, Object.assign({ seed: 9, chains: 50, lanes_per_chain: 2 }, extra));
  const a = mk({}), b = mk({ devices: [0, 0, 0] });
  a.burn(80); b.burn(80);
  const sa = a.sample(20), sb = b.sample(20);
  for (const nm of ['p1', 'n1', 'm']) assert.deepStrictEqual(Array.from(sa[nm]), Array.from(sb[nm]), nm);
  assert.deepStrictEqual(Array.from(a.info().steppers.m.accepts), Array.from(b.info().steppers.m.accepts));
  // convergence diagnostics and per-chain starts through the front-end
  const conv = a.convergence();
  assert.ok(conv.p1.rhat[0] > 0.9 && isFinite(conv.n1.rhat[0]) && conv.p1.ess[0] > 10 && conv.m.ess[0] > 0, JSON.stringify(conv));   // 100 steps of a bimodal model: not converged, only sanity
  assert.strictEqual(a.sample_on_device(30), 30);          // draws stay in HBM, summaries come from there
  assert.ok(a.moments().p1.mean[0] > 0 && a.moments().p1.mean[0] < 1);
  const qs = a.quantiles([0.025, 0.5, 0.975]);
  assert.ok(qs.p1[0][0] <= qs.p1[0][1] && qs.p1[0][1] <= qs.p1[0][2] && qs.m[0][2] <= 1 && qs.n1[0][0] >= 1, JSON.stringify(qs));
  a.init_chains((c) => ({ p1: 0.1 + 0.8 * (c / 50), n1: 1 + (c % 5), m: c % 2 }));
  const st = a.state;
  assert.strictEqual(st.n1[7], 3); assert.strictEqual(st.m[7], 1); assert.ok(Math.abs(st.p1[25] - 0.5) < 1e-12);
  a.burn(5);
  a.close(); b.close();
  const mb = um.build('multi_bern');
  assert.throws(() => new mcmc.AmwgSampler(mb.params, mb.log_post, mb.data, { seed: 1, lanes_per_chain: 4 }), /lanes|geometry/);
}
console.log('gpu user models ok');
