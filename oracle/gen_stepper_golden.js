'use strict';
/*
 * gen_stepper_golden.js -- TEST INFRASTRUCTURE ONLY.
 * Runs tests/js/stepper_cases.js with the UNMODIFIED reference module (REF_DIR/mcmc.js, distributions.js): every stepper gets
 * its own Philox stream (seed SEED, chain = the case's stream id), installed as Math.random around each of its calls.
 *   node oracle/gen_stepper_golden.js > tests/golden/steppers.json
 */
const path = require('path');
const REF_DIR = require('./ref_dir.js').refDir() || '/root/reference';
const mcmc = require(path.join(REF_DIR, 'mcmc.js'));
const ld = require(path.join(REF_DIR, 'distributions.js'));
const { stream } = require('./philox.js');
const SEED = 20260926;

function make(Class, params, state, log_post, options, streamId) {
  const rand = stream(SEED, streamId);
  const st = new Class(params, state, log_post, options);
  const wrapped = {};
  for (const m of ['step', 'info', 'start_adaptation', 'stop_adaptation']) {
    wrapped[m] = function () {
      const saved = Math.random;
      Math.random = rand;
      try { return st[m].apply(st, arguments); } finally { Math.random = saved; wrapped.uniforms = rand.count || 0; }
    };
  }
  return wrapped;
}

// JSON has no Infinity/NaN/-0: numbers travel as hex strings of their bits
function enc(v) {
  if (typeof v === 'number') { const b = Buffer.alloc(8); b.writeDoubleBE(v); return 'f64:' + b.toString('hex'); }
  if (Array.isArray(v)) return v.map(enc);
  if (v && typeof v === 'object') { const o = {}; for (const k of Object.keys(v)) o[k] = enc(v[k]); return o; }
  return v;
}

const cases = require('../tests/js/stepper_cases.js')({ mcmc, ld, make });
const out = { seed: SEED, cases: {} };
for (const name of Object.keys(cases)) out.cases[name] = enc(cases[name]());
process.stdout.write(JSON.stringify(out));
