'use strict';
// translate_steppers_cli.js -- test helper (no GPU): runs tests/js/stepper_cases.js against a recording stand-in for the stepper
// classes.  For every stepper a scenario creates, the PRODUCT's translator turns its zero-argument log_post into HIP text exactly as
// the stepper constructor would (state object recognised by identity, fixed entries after the stepped ones), and the closure itself
// is evaluated at perturbed states to give the expected values.   node tests/js/translate_steppers_cli.js <outdir>
// writes <outdir>/stepper_<case>_<k>.{hip,arrays.bin,meta.json,states.json}
const fs = require('fs');
const path = require('path');
const { mcmc, ld } = require('../../bayes.js_amd');
global.ld = ld;
const out = process.argv[2];
const names = [];

function flat(v, o) { if (Array.isArray(v)) v.forEach((e) => flat(e, o)); else o.push(Number(v)); return o; }
function shape(v) { return Array.isArray(v) ? [v.length].concat(Array.isArray(v[0]) ? shape(v[0]) : []) : [1]; }
function setFlat(target, key, vals, off) {       // writes vals[off...] into target[key] in place, returns the next offset
  if (!Array.isArray(target[key])) { target[key] = vals[off]; return off + 1; }
  const rec = (a) => { for (let i = 0; i < a.length; i++) { if (Array.isArray(a[i])) rec(a[i]); else a[i] = vals[off++]; } };
  rec(target[key]);
  return off;
}
const bits = (v) => { const b = Buffer.alloc(8); b.writeDoubleBE(v); return b.toString('hex'); };

let current = '', count = 0;
function make(Class, params, state, log_post, options, streamId, free) {
  const name = 'stepper_' + current + '_' + (count++);
  const stepped = Object.keys(params);
  const isNum = (v) => typeof v === 'number' || (Array.isArray(v) && flat(v, []).every((e) => typeof e === 'number' && e === e || typeof e === 'number'));
  const fixed = Object.keys(state).filter((k) => stepped.indexOf(k) < 0 && isNum(state[k]) && typeof state[k] !== 'string');
  const keys = stepped.concat(fixed);
  const layoutParams = {};
  for (const k of keys) layoutParams[k] = { dim: shape(state[k]) };
  const tr = mcmc.translate(log_post, layoutParams, undefined, Object.assign({ state_object: state }, free));
  fs.writeFileSync(path.join(out, name + '.hip'), tr.source);
  let bytes = 4;
  for (const a of tr.arrays) bytes += 8 + a.length * 8;
  const buf = Buffer.alloc(bytes);
  let o = 0;
  buf.writeUInt32LE(tr.arrays.length, o); o += 4;
  for (const a of tr.arrays) {
    buf.writeBigUInt64LE(BigInt(a.length), o); o += 8;
    for (let i = 0; i < a.length; i++) { buf.writeDoubleLE(a[i], o); o += 8; }
  }
  fs.writeFileSync(path.join(out, name + '.arrays.bin'), buf);
  fs.writeFileSync(path.join(out, name + '.meta.json'), JSON.stringify({ name, P: tr.P, derived: tr.derived, lds_bytes: tr.lds_bytes, lds_bytes_one_lane: tr.lds_bytes_one_lane,
    parallel: tr.parallel, max_threads: tr.max_threads, work_per_eval: tr.work_per_eval, work_one_lane: tr.work_one_lane, rows_n_obs: tr.rows_n_obs, rows_groups: tr.rows_groups, rows_sweep: tr.rows_sweep, cert_tail_n: tr.cert_tail_n, rows_cert: tr.rows_cert, array_keys: tr.array_keys, array_types: tr.array_types, array_len: tr.arrays.map((a) => a.length),
    keys }));
  // expected values: the closure at the initial state and at perturbed states (integers stay integers, 0/1 entries flip)
  const saved = JSON.stringify(keys.map((k) => state[k]));
  const init = flat(keys.map((k) => state[k]), []);
  const pts = [];
  let seed = 12345 + streamId;
  const rnd = () => { seed = (seed * 1103515245 + 12345) % 2147483648; return seed / 2147483648; };
  const binary = /Binary/.test(Class.tag || '') || (params[stepped[0]] && params[stepped[0]].type === 'binary' && stepped.length === 1);
  for (let t = 0; t < 24; t++) {
    const vals = init.map((v, i) => {
      if (t === 0) return v;
      if (binary && i < flat(stepped.map((k) => state[k]), []).length) return rnd() < 0.5 ? 0 : 1;
      if (Number.isInteger(v)) return Math.max(0, v + Math.floor(rnd() * 7) - 3);
      return v + (rnd() - 0.5) * (1 + Math.abs(v));
    });
    let off = 0;
    for (const k of keys) off = setFlat(state, k, vals, off);
    const lp = log_post();
    pts.push({ state: vals.map(bits), lp: bits(lp), derived: tr.derived.map((d) => bits(state[d])) });
  }
  const back = JSON.parse(saved);
  keys.forEach((k, i) => { if (Array.isArray(state[k])) setFlat(state, k, flat(back[i], []), 0); else state[k] = back[i]; });
  for (const d of tr.derived) delete state[d];
  fs.writeFileSync(path.join(out, name + '.states.json'), JSON.stringify(pts));
  names.push(name);
  const self = { step() { return Array.isArray(state[stepped[0]]) ? state[stepped[0]] : (stepped.length === 1 ? state[stepped[0]] : state); }, info() { return {}; }, start_adaptation() {}, stop_adaptation() {} };
  return self;
}

const classes = {};
for (const c of ['RealMetropolisStepper', 'IntMetropolisStepper', 'MultiRealComponentMetropolisStepper', 'MultiIntComponentMetropolisStepper', 'BinaryStepper', 'BinaryComponentStepper', 'AmwgStepper']) classes[c] = { tag: c };
const shim = Object.assign({}, mcmc, classes);
const cases = require('./stepper_cases.js')({ mcmc: shim, ld, make });
for (const name of Object.keys(cases)) { current = name; count = 0; cases[name](); }
fs.writeFileSync(path.join(out, 'steppers.index.json'), JSON.stringify(names));
console.log(names.join(' '));
