#!/usr/bin/env node
'use strict';
/*
 * bench/ref_cpu.js -- the CPU baseline leg of bench.py (BASELINE.md section 3, SURVEY.md section 8d): the UNMODIFIED reference
 * sampler (`require(<ref>/mcmc.js)`, `<ref>/distributions.js`; mcmc.js:1035-1039 `burn`, README.md:252) on ONE host core, on the
 * same seeded synthetic model + data as the GPU run.  Not part of the product: it only times the reference.
 *
 *   node bench/ref_cpu.js [--workload cfg2|cfg3|cfg4|cfg5|readme] [--ref DIR] [--budget SECONDS]
 *
 * The reference directory is --ref, else oracle/ref_dir.js's search ($AMWG_REF_DIR, /root/reference, oracle/_ref = the copy `make -C oracle ref`
 * puts beside the oracle so that it travels to the GPU box); its two files are hashed against oracle/ref.sha256 (`unmodified`).  If none exists the script prints
 * {"available": false} and exits 0 (the GPU box has no copy of the reference; bench.py then times the C port instead and says so).
 * Method (BASELINE.md section 3.2): one throw-away warm-up of >= 2000 sampler-steps or 2 s (V8's JIT), then the MEDIAN of 5 timed
 * burn(n) repeats, n sized from the warm-up rate so that each repeat takes >= 1 s; timer process.hrtime.bigint().
 * Data: oracle/synth.js (test infrastructure; the same Philox recipe the Python harness uses), models oracle/ref_models.js.
 */
const fs = require('fs'), path = require('path');
const args = process.argv.slice(2);
function opt(name, dflt) { const i = args.indexOf('--' + name); return i >= 0 ? args[i + 1] : dflt; }
const workload = opt('workload', 'cfg2');
const refDir = path.resolve(opt('ref', require('../oracle/ref_dir.js').refDir() || '/root/reference'));
const budget = parseFloat(opt('budget', '1.0'));     // seconds per timed repeat
if (!fs.existsSync(path.join(refDir, 'mcmc.js'))) { console.log(JSON.stringify({ available: false, ref_dir: refDir })); process.exit(0); }
const mcmc = require(path.join(refDir, 'mcmc.js')), ld = require(path.join(refDir, 'distributions.js'));
const synth = require('../oracle/synth.js'), models = require('../oracle/ref_models.js')(ld);
const DATA_SEED = 20260925;
const W = { cfg2: ['normal', 10000], readme: ['normal', 1000], cfg3: ['beta_bern', 100000], cfg4: ['hier_normal', 10000], cfg5: ['pois_glm', 50000] }[workload];
if (!W) { console.error('unknown workload ' + workload); process.exit(2); }
const fam = W[0], N = W[1];
const data = fam === 'normal' ? synth.normal(N, DATA_SEED) : fam === 'beta_bern' ? synth.bern(N, DATA_SEED) : fam === 'hier_normal' ? synth.hier(N, 32, DATA_SEED) : synth.glm(N, DATA_SEED);
const m = models[fam];
const params = m.params(data);
const sampler = new mcmc.AmwgSampler(params, m.log_post, data);
let P = 0;
Object.keys(params).forEach((k) => { const d = params[k].dim; P += d ? [].concat(d).reduce((a, b) => a * b, 1) : 1; });
let bestRate = 0;                       // sampler-steps/s, the fastest seen so far (the JIT keeps speeding the closure up)
function timed(n) { const t0 = process.hrtime.bigint(); sampler.burn(n); const t = Number(process.hrtime.bigint() - t0) * 1e-9; bestRate = Math.max(bestRate, n / Math.max(t, 1e-9)); return t; }
// warm-up: V8 tiers the closure up during the first few hundred evaluations
let warmSteps = 0, warmS = 0, n = 5;
while (warmSteps < 2000 && warmS < 2.0) { warmS += timed(n); warmSteps += n; n = Math.min(2000, n * 2); }
let steps = 0, secs = [], med = 0;
for (let attempt = 0; attempt < 3; attempt++) {      // size a repeat from the best rate seen; again if the median still came out short
  steps = Math.max(5, Math.ceil(bestRate * budget * 1.05));
  secs = [];
  for (let r = 0; r < 5; r++) secs.push(timed(steps));
  secs.sort((a, b) => a - b);
  med = secs[2];
  if (med >= 0.9 * budget) break;
}
console.log(JSON.stringify({
  available: true, workload: workload, model: fam, n_obs: N, components: P, steps_per_repeat: steps, repeats_s: secs, median_s: med,
  value: steps * P / med, unit: 'param-updates/s', cores: 1, node: process.version, ref_dir: refDir, warmup_steps: warmSteps,
  unmodified: require('../oracle/ref_dir.js').unmodified(refDir),
  reference_algorithmic_bytes_per_update: 2 * N * 8,
}));
