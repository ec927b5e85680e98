'use strict';
// test_gpu_steppers.js (-m gpu) -- the stand-alone stepper classes of bayes.js_amd against the UNMODIFIED reference's classes
// (tests/golden/steppers.json, oracle/gen_stepper_golden.js: same scenarios, every stepper on its own Philox stream).
// Bit for bit: every value step() returned, the final state object, info() incl. the stop/start_adaptation sequence.
const assert = require('assert');
const path = require('path');
const { mcmc, ld } = require('../../bayes.js_amd');
const gold = require(path.join('..', 'golden', 'steppers.json'));
global.ld = ld;

function dec(v) {
  if (typeof v === 'string' && v.startsWith('f64:')) return Buffer.from(v.slice(4), 'hex').readDoubleBE(0);
  if (Array.isArray(v)) return v.map(dec);
  if (v && typeof v === 'object') { const o = {}; for (const k of Object.keys(v)) o[k] = dec(v[k]); return o; }
  return v;
}
function same(a, b, where) {
  if (typeof a === 'number' || typeof b === 'number') { assert.ok(Object.is(a, b), where + ': ' + a + ' vs ' + b); return; }
  if (Array.isArray(a) || Array.isArray(b)) {
    assert.ok(Array.isArray(a) && Array.isArray(b) && a.length === b.length, where + ': array shape');
    for (let i = 0; i < a.length; i++) same(a[i], b[i], where + '[' + i + ']');
    return;
  }
  if (a && typeof a === 'object') {
    assert.deepStrictEqual(Object.keys(a).sort(), Object.keys(b).sort(), where + ': keys');
    for (const k of Object.keys(a)) same(a[k], b[k], where + '.' + k);
    return;
  }
  assert.strictEqual(a, b, where);
}

const made = [];
function make(Class, params, state, log_post, options, streamId, free) {
  const st = new Class(params, state, log_post, Object.assign({}, options, free, { seed: gold.seed, chain_offset: streamId }));
  made.push(st);
  return st;
}
const cases = require('./stepper_cases.js')({ mcmc, ld, make });

for (const name of Object.keys(cases)) {
  const want = dec(gold.cases[name]);
  const got = cases[name]();
  if (name === 'amwg_normal') {
    // derived key: refreshed at the final state of each step here; the reference leaves the value of the last evaluated proposal
    for (let i = 0; i < got.ret.length; i++) { assert.ok(Object.is(got.ret[i][2], got.ret[i][1] * got.ret[i][1])); got.ret[i].pop(); want.ret[i].pop(); }
    assert.ok(Object.is(got.state.var, got.state.sigma * got.state.sigma));
    delete got.state.var; delete want.state.var;
    // info(): the reference labels entry i with param_names[i] although the sub-steppers were shuffled (mcmc.js:887 vs :909)
    const bag = (o) => Object.keys(o).map((k) => JSON.stringify(o[k], Object.keys(o[k]).sort())).sort();
    assert.deepStrictEqual(bag(got.info), bag(want.info), name + ': info');
    delete got.info; delete want.info;
  }
  same(got, want, name);
  console.log('  ' + name + ': ' + want.ret.length + ' steps identical');
}

// steps(n): n steps in one launch == n calls of step()
{
  const s1 = { x: 0 }, s2 = { x: 0 };
  const d1 = function () { return ld.norm(s1.x, 10, 5); }, d2 = function () { return ld.norm(s2.x, 10, 5); };
  const a = new mcmc.RealMetropolisStepper({ x: { lower: -Infinity, upper: Infinity, dim: [1] } }, s1, d1, { seed: 5, constants: { s1 } });
  const b = new mcmc.RealMetropolisStepper({ x: { lower: -Infinity, upper: Infinity, dim: [1] } }, s2, d2, { seed: 5, constants: { s2 } });
  let last;
  for (let i = 0; i < 120; i++) last = a.step();
  assert.ok(Object.is(b.steps(120), last) && Object.is(s1.x, s2.x));
  assert.deepStrictEqual(a.info(), b.info());
  made.push(a, b);
}
// the reference's distributional checks (tests/test_mcmc_js.R:55-142), restated with fixed seeds and wide margins
{
  const mean = (a) => a.reduce((x, y) => x + y, 0) / a.length;
  const variance = (a) => { const m = mean(a); return a.reduce((x, y) => x + (y - m) * (y - m), 0) / (a.length - 1); };
  const every = (a, k) => a.filter((_, i) => i % k === 0);
  {   // RealMetropolisStepper on Normal(10, 5)
    const state = { x: 0 }, posterior = function () { return ld.norm(state.x, 10, 5); };
    const st = new mcmc.RealMetropolisStepper({ x: { lower: -Infinity, upper: Infinity, dim: [1] } }, state, posterior, { seed: 101, constants: { state } });
    st.steps(2000);
    const xs = []; for (let i = 0; i < 10000; i++) xs.push(st.step());
    assert.ok(Math.abs(mean(xs) - 10) < 0.6 && Math.abs(Math.sqrt(variance(xs)) - 5) < 0.6, 'Normal(10,5): ' + mean(xs) + ' ' + Math.sqrt(variance(xs)));
    made.push(st);
  }
  {   // IntMetropolisStepper on Poisson(10)
    const state = { x: 1 }, posterior = function () { return ld.pois(state.x, 10); };
    const st = new mcmc.IntMetropolisStepper({ x: { lower: 0, upper: Infinity, dim: [1] } }, state, posterior, { seed: 102, constants: { state } });
    st.steps(2000);
    const xs = []; for (let i = 0; i < 10000; i++) xs.push(st.step());
    assert.ok(xs.every((v) => Number.isInteger(v) && v >= 0));
    assert.ok(Math.abs(mean(xs) - 10) < 0.4 && Math.abs(variance(xs) - 10) < 2.0, 'Poisson(10): ' + mean(xs) + ' ' + variance(xs));
    made.push(st);
  }
  {   // BinaryStepper on Bernoulli(0.85): independent draws
    const state = { x: 0 }, posterior = function () { return ld.bern(state.x, 0.85); };
    const st = new mcmc.BinaryStepper({ x: { type: 'binary' } }, state, posterior, { seed: 103, constants: { state } });
    const xs = []; for (let i = 0; i < 4000; i++) xs.push(st.step());
    assert.ok(Math.abs(mean(xs) - 0.85) < 0.03, 'Bernoulli(0.85): ' + mean(xs));
    made.push(st);
  }
  {   // BinaryComponentStepper: P(x1 = 1) = (0.85 + 0.15) / (0.85 + 3 * 0.15), P(x4 = 1) = (0.75 + 0.25) / (0.75 + 3 * 0.25)
    const state = { x: [[0, 0], [0, 0]] };
    const posterior = function () { return Math.log(state.x[0][0] * state.x[0][1] * 0.85 + (1 - state.x[0][0] * state.x[0][1]) * 0.15) + Math.log(state.x[1][0] * state.x[1][1] * 0.75 + (1 - state.x[1][0] * state.x[1][1]) * 0.25); };
    const st = new mcmc.BinaryComponentStepper({ x: { type: 'binary', dim: [2, 2] } }, state, posterior, { seed: 104, constants: { state } });
    st.steps(200);
    const x1 = [], x4 = []; for (let i = 0; i < 6000; i++) { const v = st.step(); x1.push(v[0][0]); x4.push(v[1][1]); }
    assert.ok(Math.abs(mean(every(x1, 3)) - 1.0 / 1.3) < 0.05 && Math.abs(mean(every(x4, 3)) - 1.0 / 1.5) < 0.05, 'multi-Bernoulli: ' + mean(x1) + ' ' + mean(x4));
    made.push(st);
  }
  console.log('  distributional checks ok');
}
made.forEach((s) => s.close());
console.log('gpu steppers ok');
