'use strict';
/*
 * gen_moments_golden.js -- TEST INFRASTRUCTURE ONLY.
 * Posterior moments of LONG runs of the unmodified reference (mcmc.js + distributions.js under the seeded Philox stream), for the
 * two-sample moment tests of the many-lane geometries (tests/test_gpu_moments.py): per chain the mean and standard deviation of every
 * component over sample(n) after burn(b), the accept / in-bounds counts, the uniforms consumed and the final state -- so the same
 * fixture serves both the statistical comparison (pooled GPU chains vs pooled reference chains) and the decision-for-decision one
 * (GPU chain ids 0..7 at 64 lanes vs these eight runs).
 *     node oracle/gen_moments_golden.js         -> tests/golden/moments_<case>.json
 */
const fs = require('fs');
const path = require('path');
const h = require('./ref_harness.js');
const OUT = path.join(__dirname, '..', 'tests', 'golden');

const CASES = [
  // a Poisson GLM the reference converges on: 8 real coefficients + the int change point (SURVEY.md section 8(d) cfg5 at N = 500)
  { name: 'moments_glm_n500', model: 'pois_glm', N: 500, data_seed: 20260925, seed: 20260925, chains: [0, 1, 2, 3, 4, 5, 6, 7], burn: 5000, sample: 15000 },
  // the hierarchical family at N = 640 in 8 groups (cfg4's structure)
  { name: 'moments_hier_n640', model: 'hier_normal', N: 640, G: 8, data_seed: 20260925, seed: 20260925, chains: [0, 1, 2, 3, 4, 5, 6, 7], burn: 3000, sample: 12000 },
];

const want = process.argv.slice(2);
for (const c of CASES) {
  if (want.length && want.indexOf(c.name) < 0) continue;
  const t0 = Date.now();
  const cc = Object.assign({}, c, { schedule: [{ op: 'burn', n: c.burn }, { op: 'sample', n: c.sample, keep: 0 }] });
  const data = h.makeData(cc);
  const chains = [];
  for (const ch of c.chains) {
    // the harness keeps running sums only; the second moment needs the draws: run the chain with every draw kept, reduce here
    const r = h.runChain(Object.assign({}, cc, { schedule: [{ op: 'burn', n: c.burn }, { op: 'sample', n: c.sample }] }), data, ch);
    const rows = r.samples[0].draws, P = rows[0].length, n = rows.length;
    const mean = new Array(P).fill(0), sd = new Array(P).fill(0);
    for (let t = 0; t < n; t++) for (let j = 0; j < P; j++) mean[j] += rows[t][j];
    for (let j = 0; j < P; j++) mean[j] /= n;
    for (let t = 0; t < n; t++) for (let j = 0; j < P; j++) { const d = rows[t][j] - mean[j]; sd[j] += d * d; }
    for (let j = 0; j < P; j++) sd[j] = Math.sqrt(sd[j] / (n - 1));
    chains.push({ chain: ch, mean: mean, sd: sd, kept: n, accepts: r.accepts, inbounds: r.inbounds, uniforms: r.uniforms, final_state: r.final_state,
                  prop_log_scale: r.prop_log_scale, batch_count: r.batch_count, comp_opts: r.comp_opts, params_completed: r.params_completed });
  }
  fs.writeFileSync(path.join(OUT, c.name + '.json'), h.stringify({ case: c, chains: chains }));
  console.log(c.name, ((Date.now() - t0) / 1000).toFixed(1) + 's');
}
