'use strict';
/*
 * gen_user_golden.js -- TEST INFRASTRUCTURE ONLY.
 * Runs the user-style closures of tests/js/user_models.js through the UNMODIFIED reference sampler
 * (/root/reference, build container only) under the seeded Philox Math.random, and records
 *   - whole trajectories (draws incl. derived quantities, accept/flip counts, adaptation state), as oracle/gen_golden.js does
 *   - log_post values at 40 visited + perturbed states (the translator is checked against these without a GPU)
 * into tests/golden/user_<name>.json.     node oracle/gen_user_golden.js [name ...]
 */
const fs = require('fs');
const path = require('path');
const h = require('./ref_harness.js');
const um = require('../tests/js/user_models.js');
const OUT = path.join(__dirname, '..', 'tests', 'golden');
const SEED = 20260925;
global.ld = h.ld;   // the closures are written against a global `ld`, like README.md's browser examples

function flat(v) { const o = []; (function r(x) { if (Array.isArray(x)) x.forEach(r); else o.push(x); })(v); return o; }
function nestLike(shape, values, pos) {   // values -> nested array shaped like `shape`
  if (!Array.isArray(shape)) return values[pos.i++];
  return shape.map((e) => nestLike(e, values, pos));
}

const want = process.argv.slice(2);
for (const name of um.names) {
  if (want.length && want.indexOf(name) < 0) continue;
  const t0 = Date.now();
  const m = um.build(name, SEED);
  for (const k of Object.keys(m.helpers || {})) global[k] = m.helpers[k];
  for (const k of Object.keys(m.constants || {})) global[k] = m.constants[k];
  const c = { name: 'user_' + name, model: name, seed: SEED, chains: m.chains, schedule: m.schedule, options: m.options };
  const model = { params: () => m.params, log_post: m.log_post };
  const res = { case: c, chains: m.chains.map((ch) => h.runChain(c, m.data, ch, model)) };
  // log_post at states the chains visited, and at perturbations of them (bounds violations included)
  const completed = h.mcmc.complete_params(m.params, h.mcmc.param_init_fixed);
  const names = Object.keys(completed);
  const rnd = um.lcg(777);
  const states = [];
  const P = res.chains[0].final_state.length;
  const rows = [flat(names.map((n) => completed[n].init))];
  for (const ch of res.chains) { rows.push(ch.final_state); for (const sg of ch.samples) for (let t = 0; t < sg.draws.length; t += 7) rows.push(sg.draws[t].slice(0, P)); }
  for (let r = 0; r < rows.length && states.length < 40; r++) {
    states.push(rows[r]);
    const pert = rows[r].map((v, j) => { let ci = 0; for (const n of names) { const len = flat(completed[n].init).length; if (j < ci + len) { const ty = completed[n].type; return ty === 'real' ? v + (rnd() - 0.5) * 0.3 : (ty === 'int' ? v + Math.round((rnd() - 0.5) * 3) : (rnd() < 0.5 ? 0 : 1)); } ci += len; } return v; });
    states.push(pert);
  }
  res.log_post_checks = states.map((vals) => {
    const st = {}; const pos = { i: 0 };
    for (const n of names) st[n] = nestLike(completed[n].init, vals, pos);
    const lp = m.log_post(st, m.data);
    const derived = Object.keys(st).filter((k) => names.indexOf(k) < 0).map((k) => st[k]);
    return { state: vals, log_post: lp, derived };
  });
  fs.writeFileSync(path.join(OUT, 'user_' + name + '.json'), h.stringify(res));
  console.log(name, ((Date.now() - t0) / 1000).toFixed(1) + 's', 'states:', res.log_post_checks.length);
}
