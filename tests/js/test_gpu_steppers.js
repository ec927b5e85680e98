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
made.forEach((s) => s.close());
console.log('gpu steppers ok');
