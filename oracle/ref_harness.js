'use strict';
/*
 * ref_harness.js -- TEST INFRASTRUCTURE ONLY.
 * Runs the UNMODIFIED reference sampler (require(REF_DIR/mcmc.js), default
 * /root/reference) with Math.random replaced by the Philox twin, and records
 * everything the parity tests compare against: every draw, per-component
 * accept decisions, final adaptation state, uniforms consumed.
 *
 *   node oracle/ref_harness.js <case.json>   -> JSON on stdout
 *   require('./ref_harness.js').runCase(caseObj)
 *
 * Accept decisions are observed, not re-derived: each one-dimensional stepper
 * instance gets its log_post field (mcmc.js:436) wrapped to capture the two
 * densities of mcmc.js:524-526, and the accept test of mcmc.js:527-528 is
 * replayed with the same Math.exp and the uniform the stream just handed out.
 */
const path = require('path');
const REF_DIR = require('./ref_dir.js').refDir() || '/root/reference';
const mcmc = require(path.join(REF_DIR, 'mcmc.js'));
const ld = require(path.join(REF_DIR, 'distributions.js'));
const { stream } = require('./philox.js');
const synth = require('./synth.js');
const makeModels = require('./ref_models.js');
const models = makeModels(ld);

function makeData(c) {
  switch (c.model) {
    case 'normal': return c.data ? { x: c.data.x } : synth.normal(c.N, c.data_seed);
    case 'beta_bern': return synth.bern(c.N, c.data_seed);
    case 'hier_normal': return synth.hier(c.N, c.G || 32, c.data_seed);
    case 'pois_glm': return synth.glm(c.N, c.data_seed);
  }
  throw new Error('unknown model ' + c.model);
}

function flattenSteppers(amwg) {
  const out = [];
  const rec = (s) => { if (Array.isArray(s)) s.forEach(rec); else out.push(s); };
  for (const s of amwg.substeppers) { if (s.substeppers) rec(s.substeppers); else out.push(s); }
  return out;
}
function flat(v) { const o = []; const rec = (x) => { if (Array.isArray(x)) x.forEach(rec); else o.push(x); }; rec(v); return o; }

function runChain(c, data, chain, model) {
  const rand = stream(c.seed, chain);
  const saved = Math.random;
  Math.random = rand;
  try {
    const m = model || (c.hyper ? makeModels(ld, { [c.model]: c.hyper }) : models)[c.model];
    const params = m.params(data);
    // (a case may change a parameter's description -- type, bounds, init -- before the reference completes it: c.param_overrides = {name: {...}})
    for (const nm of Object.keys(c.param_overrides || {})) Object.assign(params[nm], c.param_overrides[nm]);
    const sampler = new mcmc.AmwgSampler(params, m.log_post, data, c.options);
    const names = Object.keys(params);
    const comps = flattenSteppers(sampler.steppers[0]);
    const accepts = comps.map(() => 0), inbounds = comps.map(() => 0);
    // merged per-component stepper options and completed params as the reference built them
    const comp_opts = comps.map((s) => ({ prop_log_scale: s.prop_log_scale, batch_size: s.batch_size,
      max_adaptation: s.max_adaptation, initial_adaptation: s.initial_adaptation,
      target_accept_rate: s.target_accept_rate, is_adapting: s.is_adapting }));
    const params_completed = names.map((nm) => { const p = sampler.params[nm];
      return { name: nm, type: p.type, dim: p.dim, lower: p.lower, upper: p.upper, init: flat(p.init) }; });
    comps.forEach((st, ci) => {
      if (st.prop_log_scale === undefined) {   // BinaryStepper (mcmc.js:740-767): count evaluations and flips
        const protoStep = Object.getPrototypeOf(st).step;
        st.step = function () {
          const before = this.state[this.param_name];
          const r = protoStep.call(this);
          inbounds[ci]++;
          if (this.state[this.param_name] !== before) accepts[ci]++;
          return r;
        };
        return;
      }
      const lp0 = st.log_post; let seen = [];
      st.log_post = function () { const v = lp0(); seen.push(v); return v; };
      const protoStep = Object.getPrototypeOf(st).step;
      st.step = function () {
        seen = [];
        const r = protoStep.call(this);
        if (seen.length === 2) { inbounds[ci]++; if (Math.exp(seen[1] - seen[0]) > rand.last) accepts[ci]++; }
        return r;
      };
    });
    const out = { chain: chain, comp_opts: comp_opts, params_completed: params_completed };
    const segs = [];
    for (const seg of c.schedule) {           // [{op:'burn',n}|{op:'sample',n,thin?}|{op:'stop'}|{op:'start'}]
      if (seg.op === 'burn') sampler.burn(seg.n);
      else if (seg.op === 'stop') sampler.stop_adaptation();
      else if (seg.op === 'start') sampler.start_adaptation();
      else if (seg.op === 'sample') {
        if (seg.thin) sampler.thin(seg.thin);
        const s = sampler.sample(seg.n);
        // every recorded key: the parameters, then the closure's derived quantities (mcmc.js:1009-1013)
        const keys = Object.keys(s);
        const kept = s[names[0]].length, rows = [];
        const keep = seg.keep === undefined ? kept : Math.min(kept, seg.keep);
        for (let t = 0; t < keep; t++) { let row = []; for (const nm of keys) row = row.concat(flat(s[nm][t])); rows.push(row); }
        // running sums over ALL kept draws so long runs can be checked without storing them
        const P = rows.length ? rows[0].length : 0, sum = new Array(P).fill(0);
        for (let t = 0; t < kept; t++) { let j = 0; for (const nm of keys) for (const v of flat(s[nm][t])) sum[j++] += v; }
        segs.push({ kept: kept, draws: rows, sum: sum, keys: keys });
      }
    }
    out.samples = segs;
    out.final_state = []; for (const nm of names) out.final_state = out.final_state.concat(flat(sampler.state[nm]));
    out.accepts = accepts; out.inbounds = inbounds;
    const orZero = (v) => (v === undefined ? 0 : v);   // BinarySteppers keep no adaptation state
    out.prop_log_scale = comps.map((s) => orZero(s.prop_log_scale));
    out.batch_count = comps.map((s) => orZero(s.batch_count));
    out.acceptance_count = comps.map((s) => orZero(s.acceptance_count));
    out.iterations_since_adaption = comps.map((s) => orZero(s.iterations_since_adaption));
    out.uniforms = rand.count;
    out.log_post = sampler.log_post();
    // the order of the named sub-steppers after the last in-place shuffle (mcmc.js:887)
    out.named_order = sampler.steppers[0].substeppers.map((s) => names.indexOf(s.param_name));
    return out;
  } finally { Math.random = saved; }
}

function runCase(c) {
  const data = makeData(c);
  const res = { case: c, chains: c.chains.map((ch) => runChain(c, data, ch)) };
  if (c.store_data) res.data = data;
  return res;
}

// JSON has no Infinity/NaN/-0: encode them as tagged strings (tests/golden_io.py decodes)
function stringify(o) {
  return JSON.stringify(o, (k, v) => (typeof v === 'number' && !isFinite(v)) ? (isNaN(v) ? '__nan' : (v > 0 ? '__inf' : '__-inf')) : (Object.is(v, -0) ? '__-0' : v));
}

module.exports = { stringify, runCase, runChain, makeData, models, mcmc, ld };

if (require.main === module) {
  const c = JSON.parse(require('fs').readFileSync(process.argv[2], 'utf8'));
  process.stdout.write(stringify(runCase(c)));
}
