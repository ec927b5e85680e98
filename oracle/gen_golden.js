'use strict';
/*
 * gen_golden.js -- TEST INFRASTRUCTURE ONLY.
 * Generates tests/golden/*.json by running the unmodified reference
 * (/root/reference, present only in the build container) under the seeded
 * Philox stream.  Run:  node oracle/gen_golden.js [case-name ...]
 * The fixtures are committed; this script is committed so they can be regenerated.
 */
const fs = require('fs');
const path = require('path');
const h = require('./ref_harness.js');
const OUT = path.join(__dirname, '..', 'tests', 'golden');
const SEED = 20260925, DSEED = 20260925;

const CASES = [
  // BASELINE.json configs[0]: README.md:20 data, burn 1000 + 5000 draws
  { name: 'cfg1_heights', model: 'normal', data: { x: [183, 192, 182, 183, 177, 185, 188, 188, 182, 185] }, store_data: true,
    seed: SEED, chains: [0, 1, 2, 3], schedule: [{ op: 'burn', n: 1000 }, { op: 'sample', n: 5000, keep: 200 }] },
  { name: 'normal_n1000', model: 'normal', N: 1000, data_seed: DSEED, store_data: true,
    seed: SEED, chains: [0, 1, 5, 77777], schedule: [{ op: 'burn', n: 300 }, { op: 'sample', n: 300, keep: 100 }] },
  // configs[1] at full size, first and last chain id of the 65 536
  { name: 'cfg2_full', model: 'normal', N: 10000, data_seed: DSEED,
    seed: SEED, chains: [0, 65535], schedule: [{ op: 'burn', n: 500 }, { op: 'sample', n: 500, keep: 20 }] },
  // option merging (mcmc.js:869-878), stop/start adaptation, thinning
  { name: 'normal_opts', model: 'normal', N: 200, data_seed: DSEED + 1, store_data: true,
    options: { batch_size: 10, target_accept_rate: 0.3, max_adaptation: 0.5, prop_log_scale: -1, params: { mu: { max_adaptation: 0.1 } } },
    seed: SEED + 1, chains: [0, 9],
    schedule: [{ op: 'burn', n: 105 }, { op: 'stop' }, { op: 'sample', n: 50 }, { op: 'start' }, { op: 'sample', n: 100, thin: 7 }] },
  // non-default prior hyper-parameters (amwg_model_desc.hyper)
  { name: 'normal_hyper', model: 'normal', N: 150, data_seed: DSEED + 2, store_data: true, hyper: [2.5, 7, 0.5, 30],
    seed: SEED + 2, chains: [0, 4], schedule: [{ op: 'burn', n: 150 }, { op: 'sample', n: 100 }] },
  { name: 'beta_bern_hyper', model: 'beta_bern', N: 300, data_seed: DSEED + 3, store_data: true, hyper: [1, 1],
    seed: SEED + 3, chains: [0], schedule: [{ op: 'burn', n: 150 }, { op: 'sample', n: 100 }] },
  { name: 'beta_bern_hyper2', model: 'beta_bern', N: 300, data_seed: DSEED + 3, store_data: true, hyper: [0.5, 3.5],
    seed: SEED + 3, chains: [1], schedule: [{ op: 'burn', n: 150 }, { op: 'sample', n: 100 }] },
  { name: 'hier_hyper', model: 'hier_normal', N: 300, G: 5, data_seed: DSEED + 4, store_data: true, hyper: [4, 20, 0, 50, 3],
    seed: SEED + 4, chains: [0], schedule: [{ op: 'burn', n: 100 }, { op: 'sample', n: 60 }] },
  { name: 'glm_hyper', model: 'pois_glm', N: 200, data_seed: DSEED + 5, store_data: true, hyper: [0.1, 2],
    seed: SEED + 5, chains: [0], schedule: [{ op: 'burn', n: 80 }, { op: 'sample', n: 40 }] },
  { name: 'beta_bern_n2000', model: 'beta_bern', N: 2000, data_seed: DSEED, store_data: true,
    seed: SEED, chains: [0, 1, 2], schedule: [{ op: 'burn', n: 400 }, { op: 'sample', n: 400, keep: 100 }] },
  { name: 'cfg3_full', model: 'beta_bern', N: 100000, data_seed: DSEED,
    seed: SEED, chains: [0, 262143], schedule: [{ op: 'burn', n: 150 }, { op: 'sample', n: 150, keep: 20 }] },
  { name: 'hier_small', model: 'hier_normal', N: 640, G: 8, data_seed: DSEED, store_data: true,
    seed: SEED, chains: [0, 1], schedule: [{ op: 'burn', n: 200 }, { op: 'sample', n: 200, keep: 50 }] },
  { name: 'cfg4_full', model: 'hier_normal', N: 10000, G: 32, data_seed: DSEED,
    seed: SEED, chains: [0, 16383], schedule: [{ op: 'burn', n: 150 }, { op: 'sample', n: 150, keep: 10 }] },
  // configs[3] at full size with a BOUNDED and with an INTEGER theta (mcmc.js:520-522: a proposal outside draws no accept uniform; mcmc.js:597: rounded proposals):
  // what the sweep kernel's walk over the stream has to get right when not every update draws three uniforms
  { name: 'cfg4_theta_bounded', model: 'hier_normal', N: 10000, G: 32, data_seed: DSEED, param_overrides: { theta: { lower: 2.5, upper: 7.5, init: 5 } },
    seed: SEED, chains: [0, 16383], schedule: [{ op: 'burn', n: 150 }, { op: 'sample', n: 150, keep: 10 }] },
  { name: 'cfg4_theta_int', model: 'hier_normal', N: 10000, G: 32, data_seed: DSEED, param_overrides: { theta: { type: 'int', init: 5 } },
    seed: SEED, chains: [0, 16383], schedule: [{ op: 'burn', n: 150 }, { op: 'sample', n: 150, keep: 10 }] },
  { name: 'glm_small', model: 'pois_glm', N: 500, data_seed: DSEED, store_data: true,
    seed: SEED, chains: [0, 1], schedule: [{ op: 'burn', n: 150 }, { op: 'sample', n: 150, keep: 50 }] },
  { name: 'cfg5_full', model: 'pois_glm', N: 50000, data_seed: DSEED,
    seed: SEED, chains: [0, 65535], schedule: [{ op: 'burn', n: 150 }, { op: 'sample', n: 150, keep: 10 }] },
];

function checksum(data) { // order-sensitive sum so the Python twin of synth.js can be checked at full N
  const out = {};
  for (const k of Object.keys(data)) if (Array.isArray(data[k])) { let s = 0; for (let i = 0; i < data[k].length; i++) s += data[k][i] * (1 + (i % 7)); out[k] = s; }
  return out;
}

const want = process.argv.slice(2);
for (const c of CASES) {
  if (want.length && want.indexOf(c.name) < 0) continue;
  const t0 = Date.now();
  const res = h.runCase(c);
  res.data_checksum = checksum(h.makeData(c));
  fs.writeFileSync(path.join(OUT, c.name + '.json'), h.stringify(res));
  console.log(c.name, ((Date.now() - t0) / 1000).toFixed(1) + 's');
}
