'use strict';
/*
 * mcmc.js -- JavaScript front-end of the MI355X many-chain AMWG sampler.
 *
 * Keeps the user-facing surface of the reference's sampler (SURVEY.md §8b):
 *     new mcmc.AmwgSampler(params, log_post, data, options)     mcmc.js:1090-1099, 940-966
 *     .burn(n) .sample(n) .step() .thin(k) .monitor(names)      mcmc.js:1035, 1005, 985, 1053, 1045
 *     .start_adaptation() .stop_adaptation() .info() .state     mcmc.js:1060-1073, 977, 964
 *     mcmc.complete_params, mcmc.param_init_fixed               mcmc.js:357-403, 313-341
 * and runs the stepping on the GPU through the N-API shim (csrc/amwg_napi.c) over the C ABI
 * (include/amwg.h).  There is no JavaScript stepping path here: if the addon or a GPU is
 * missing, construction throws.
 *
 * log_post.  The four BASELINE.json model families have hand-tuned kernels (models.js recognises
 * the README programs from their source, or take mcmc.models.*()).  Every other closure is
 * translated to HIP by translate.js and compiled at construction with hiprtc: real, int and binary
 * parameters, any dim, every scalar ld.* density, derived quantities (`state.key = ...`).
 * options.translate = true forces the translated path for a recognised closure as well;
 * options.constants / options.helpers make free variables / helper functions of the closure
 * visible to the translator.
 *
 * What is new relative to the reference (all optional):
 *     options.chains   number of independent chains (default 1)
 *     options.seed     Philox key, number or BigInt (default: drawn from Math.random, like an unseeded run)
 *     options.devices  HIP device ordinals to shard the chains over (default [0])
 *     options.lanes_per_chain / block_threads / steps_per_launch / exact_division / group_local / full_evaluation  -> amwg_options
 * With chains === 1 every return value has the reference's shape.  With chains > 1 each
 * monitored parameter is a Float64Array laid out [draw][element][chain] with a non-enumerable
 * `.layout = {kept, len, chains, dim}`.
 */
const path = require('path');
const models = require('./models.js');
const ld = require('./ld.js');
const translator = require('./translate.js');

let nativeCache = null;
function native() {
  if (!nativeCache) {
    try {
      nativeCache = require(path.join(__dirname, 'csrc', 'amwg_napi.node'));
    } catch (e) {
      throw 'AmwgSampler (MI355X): the native addon csrc/amwg_napi.node could not be loaded (' + e.message +
            '); build it with `make -C bayes.js_amd/csrc`. There is no JavaScript fallback.';
    }
  }
  return nativeCache;
}

// ---------------------------------------------------------------------------------------------
// Parameter completion.  Semantics of mcmc.js:313-341 and :357-403, pinned by the reference's
// only deterministic test (tests/test_mcmc_js.R:39-46 with tests/test_data.js:9-74).
function param_init_fixed(type, lower, upper) {
  if (lower > upper) throw 'Can not initialize parameter where lower bound > upper bound';
  const noLo = lower === -Infinity, noHi = upper === Infinity;
  if (type === 'real') {
    if (noLo && noHi) return 0.5;
    if (noLo) return upper - 0.5;
    if (noHi) return lower + 0.5;
    if (lower <= upper) return (lower + upper) / 2;
  } else if (type === 'int') {
    if (noLo && noHi) return 1;
    if (noLo) return upper - 1;
    if (noHi) return lower + 1;
    if (lower <= upper) return Math.round((lower + upper) / 2);
  } else if (type === 'binary') {
    return 1;
  }
  throw 'Could not initialize parameter of type ' + type + '[' + lower + ', ' + upper + ']';
}

function cloneSpec(v) {                     // params are deep-copied, functions kept by reference (mcmc.js:358)
  if (Array.isArray(v)) return v.map(cloneSpec);
  if (v && typeof v === 'object') { const o = {}; for (const k of Object.keys(v)) o[k] = cloneSpec(v[k]); return o; }
  return v;
}
function filled(dim, init) {                // nested array of shape dim; a function is called once per element
  if (dim.length === 0) throw "create_array can't create a dimensionless array";
  const out = new Array(dim[0]);
  for (let i = 0; i < dim[0]; i++) out[i] = dim.length === 1 ? (typeof init === 'function' ? init() : init) : filled(dim.slice(1), init);
  return out;
}
function shapeOf(a) { return Array.isArray(a[0]) ? [a.length].concat(shapeOf(a[0])) : [a.length]; }
function sameShape(a, b) { return a.length === b.length && a.every((v, i) => v == b[i]); }
const isScalarDim = (dim) => sameShape(dim, [1]);
const TYPE_ID = { real: 0, int: 1, binary: 2 };   // AMWG_REAL / AMWG_INT / AMWG_BINARY

function complete_params(params_to_complete, param_init) {
  const params = cloneSpec(params_to_complete);
  for (const name of Object.keys(params)) {
    const p = params[name];
    if (!p.hasOwnProperty('type')) p.type = 'real';
    if (!p.hasOwnProperty('dim')) p.dim = [1];
    if (typeof p.dim === 'number') p.dim = [p.dim];
    if (p.type == 'binary') { p.upper = 1; p.lower = 0; }
    if (!p.hasOwnProperty('upper')) p.upper = Infinity;
    if (!p.hasOwnProperty('lower')) p.lower = -Infinity;
    if (p.hasOwnProperty('init')) {
      if (isScalarDim(p.dim) && typeof p.init === 'function') p.init = p.init();
      else if (!isScalarDim(p.dim) && !Array.isArray(p.init)) p.init = filled(p.dim, p.init);
    } else if (isScalarDim(p.dim)) {
      p.init = param_init(p.type, p.lower, p.upper);
    } else {
      p.init = filled(p.dim, () => param_init(p.type, p.lower, p.upper));
    }
  }
  return params;
}

// ---------------------------------------------------------------------------------------------
// Stepper options.  AmwgStepper merges per-parameter and global options with `||`
// (mcmc.js:869-878: falsy overrides such as 0 or false are ignored -- kept, it is observable),
// then each stepper applies defaults with get_option (mcmc.js:280-285, 500-505); multi-
// dimensional parameters accept a scalar or an array of exactly the parameter's shape
// (mcmc.js:293-303, 644-649).
const OPTION_DEFAULTS = { prop_log_scale: 0, batch_size: 50, max_adaptation: 0.33, initial_adaptation: 1.0,
  target_accept_rate: 0.44, is_adapting: true };
const OPTION_KEYS = Object.keys(OPTION_DEFAULTS);

function flatten(v, out) { if (Array.isArray(v)) v.forEach((e) => flatten(e, out)); else out.push(v); return out; }

function componentOptions(name, param, options) {
  options = options || {};
  const own = (options.params && options.params[name]) || {};
  const len = param.dim.reduce((a, b) => a * b, 1);
  const perKey = {};
  for (const key of OPTION_KEYS) {
    const merged = own[key] || options[key];
    let v = (merged !== undefined && merged !== null) ? merged : OPTION_DEFAULTS[key];
    if (isScalarDim(param.dim)) {
      perKey[key] = [v];
    } else {
      if (!Array.isArray(v)) v = filled(param.dim, v);
      if (!sameShape(shapeOf(v), param.dim))
        throw 'The option ' + key + ' is of dimension [' + shapeOf(v) + '] but should be [' + param.dim + '].';
      perKey[key] = flatten(v, []);
    }
  }
  const out = [];
  for (let e = 0; e < len; e++) {
    const o = {};
    for (const key of OPTION_KEYS) o[key] = perKey[key][e];
    out.push(o);
  }
  return out;
}

// ---------------------------------------------------------------------------------------------
function toF64(a) { return a instanceof Float64Array ? a : Float64Array.from(a); }

function buildModelDesc(recog, data, params) {
  const parts = recog.extract(data, params);
  const family = recog.family;
  const hyper = new Float64Array(8);
  recog.hyper.forEach((v, i) => { hyper[i] = v; });
  const desc = { model: models.FAMILY_ID[family], hyper, G: 0, K: 0 };
  if (family === 'pois_glm') {
    desc.y = toF64(parts.y); desc.n_obs = desc.y.length; desc.K = 7; desc.x = toF64(parts.x);
    if (desc.x.length !== desc.n_obs * 7) throw 'pois_glm: data.X must hold N x 7 values';
  } else {
    if (parts.x === undefined || parts.x === null) throw 'AmwgSampler (MI355X): the data argument does not contain the observations the model loops over';
    desc.x = toF64(parts.x); desc.n_obs = desc.x.length;
    if (family === 'hier_normal') { desc.g = Int32Array.from(parts.g); desc.G = parts.G; }
  }
  return desc;
}

function nest(flat, offset, dim) {          // row-major flat values -> nested array of shape dim
  if (dim.length === 1) return Array.prototype.slice.call(flat, offset, offset + dim[0]);
  const inner = dim.slice(1).reduce((a, b) => a * b, 1), out = [];
  for (let i = 0; i < dim[0]; i++) out.push(nest(flat, offset + i * inner, dim.slice(1)));
  return out;
}

// module-private option key: the stand-alone stepper classes (below) build their engine through the same constructor and hand it
// the state object they share with the caller, {state, fixed: [{name, dim, len, flat}]}
const SHARED = Symbol('amwg.shared_state');

function AmwgSampler(params, log_post, data, options) {
  options = options || {};
  const opt = (k, d) => (options.hasOwnProperty(k) && options[k] !== undefined && options[k] !== null) ? options[k] : d;
  const shared = options[SHARED] || null;
  this.param_names = Object.keys(params);
  this.param_init_fun = opt('param_init_fun', param_init_fixed);
  this.thin(opt('thin', 1));
  this.monitor(opt('monitor', null));
  this.options = options;
  this.data = data;
  this.params = shared ? params : complete_params(params, this.param_init_fun);   // steppers take completed params (mcmc.js:419-421)

  let recog = (shared || opt('translate', false)) ? null : models.recognise(log_post);
  // a recognised family whose parameters are declared in another order than the hand-written kernel lays them out (the reference
  // accepts any order, only the stepper order depends on it, mcmc.js:839): translate the closure like any other
  if (recog && recog.paramNames && recog.paramNames.join() !== this.param_names.join()) recog = null;
  // the hierarchical family recognised from a closure's source: its loop over the group means must cover exactly the declared components
  // and every label must name one of them -- anything else is some other model, and is translated
  if (recog && recog.dimCheck) {
    let ok = false;
    try {
      const parts = recog.extract(data, this.params), p = this.params[recog.dimCheck.name];
      ok = !!p && p.dim.length === 1 && p.dim[0] === parts.G && parts.x && parts.g && parts.x.length === parts.g.length && parts.G >= 2 &&
           Array.prototype.every.call(parts.g, (v) => Number.isInteger(v) && v >= 0 && v < parts.G);
    } catch (e) { ok = false; }
    if (!ok) recog = null;
  }
  this.model = recog ? recog.family : 'translated';

  // flatten params / init / options in Object.keys order (the stepper order of mcmc.js:839)
  const descs = [], init = [], compOpts = [];
  this._layout = [];
  let base = 0;
  for (const name of this.param_names) {
    const p = this.params[name];
    if (p.type !== 'real' && p.type !== 'int' && p.type !== 'binary')
      throw "AmwgStepper can't handle parameter " + name + ' with type ' + p.type;   // message of mcmc.js:867
    if (p.type === 'binary' && recog) throw 'AmwgSampler (MI355X): the built-in ' + recog.family + ' model has no binary parameters';
    const len = p.dim.reduce((a, b) => a * b, 1);
    descs.push({ type: TYPE_ID[p.type], len, top: p.dim[0], multidim: isScalarDim(p.dim) ? 0 : 1, lower: p.lower, upper: p.upper });
    const flatInit = flatten(p.init, []);
    if (flatInit.length !== len) throw 'parameter ' + name + ': init does not match dim [' + p.dim + ']';
    flatInit.forEach((v) => init.push(v));
    componentOptions(name, p, options).forEach((o) => compOpts.push(o));
    this._layout.push({ name, base, len, dim: p.dim, scalar: isScalarDim(p.dim) });
    base += len;
  }
  this.P_stepped = base;
  // the entries of a shared state object that no parameter of this stepper owns: state slots log_post reads, never stepped
  const translatedParams = Object.assign({}, this.params);
  if (shared) for (const f of shared.fixed) {
    descs.push({ type: 3 /* AMWG_FIXED */, len: f.len, top: f.len, multidim: 1, lower: -Infinity, upper: Infinity });
    f.flat.forEach((v) => init.push(v));
    for (let e = 0; e < f.len; e++) compOpts.push(Object.assign({}, OPTION_DEFAULTS));
    translatedParams[f.name] = { dim: f.dim };
    base += f.len;
  }
  this.P = base;
  this.chains = opt('chains', 1);
  if (!(this.chains >= 1) || Math.floor(this.chains) !== this.chains) throw 'options.chains must be a positive integer';
  this.seed = opt('seed', Math.floor(Math.random() * 9007199254740992));
  // the Philox key is an unsigned 64-bit integer: a non-negative safe integer or a BigInt below 2^64 (anything else would
  // silently become another key)
  if (!(typeof this.seed === 'bigint' ? (this.seed >= 0n && this.seed < 18446744073709551616n) : (Number.isSafeInteger(this.seed) && this.seed >= 0)))
    throw 'AmwgSampler (MI355X): options.seed must be a non-negative integer (number up to 2^53 - 1, or BigInt below 2^64), got ' + String(this.seed);
  const devices = opt('devices', [opt('device', 0)]);
  // the model: a built-in family, or the closure translated to HIP (compiled by the addon with hiprtc)
  let desc = null, user = null;
  this.derived = [];
  if (recog) desc = buildModelDesc(recog, data, this.params);
  else {
    const tr = translator.translate(log_post, translatedParams, data, { constants: options.constants, helpers: options.helpers,
      lds_budget: options.lds_budget, max_threads: options.max_threads, unroll: options.unroll, state_object: shared ? shared.state : undefined });
    this.derived = tr.derived;
    this.translation = tr;
    user = { source: tr.source, arrays: tr.arrays, array_types: tr.array_types, n_derived: tr.derived.length, lds_bytes: tr.lds_bytes, lds_bytes_one_lane: tr.lds_bytes_one_lane, parallel: tr.parallel,
             max_threads: tr.max_threads, work_per_eval: tr.work_per_eval, work_one_lane: tr.work_one_lane, rows_n_obs: tr.rows_n_obs, rows_groups: tr.rows_groups, rows_sweep: tr.rows_sweep };
  }
  this.PR = this.P + this.derived.length;   // values per recorded draw

  // contiguous shards of global chain ids, one native sampler per device (SURVEY.md §8e)
  const N = native();
  this._shards = [];
  const D = Math.min(devices.length, this.chains), per = Math.floor(this.chains / D), rem = this.chains % D;
  let offset = 0;
  let lanes = opt('lanes_per_chain', 0);
  for (let r = 0; r < D; r++) {
    const count = per + (r < rem ? 1 : 0);
    const handle = (user ? N.createUser : N.create)(user || desc, descs, Float64Array.from(init), compOpts, {
      chains: count, seed: this.seed, chain_offset: opt('chain_offset', 0) + offset, device: devices[r],
      lanes_per_chain: lanes, block_threads: opt('block_threads', 0),
      steps_per_launch: opt('steps_per_launch', 0), exact_division: opt('exact_division', 0), group_local: opt('group_local', 0) ? 1 : 0, full_evaluation: Number(opt('full_evaluation', 0)) | 0, test_bound_shift: Number(opt('test_bound_shift', 0)) | 0,
      sufficient_statistics: opt('sufficient_statistics', 0) ? 1 : 0 });
    this._shards.push({ handle, offset, count, device: devices[r] });
    // one summation order for the whole job: what the first shard picked (cost model, or the measurement of lanes_per_chain: -2)
    // is what the other shards get -- a chain's draws must not depend on the shard it landed in
    if (r === 0 && D > 1 && lanes <= 0) lanes = N.launchInfo(handle).lanes_per_chain;
    offset += count;
  }
  this._host_log_post = (st) => log_post(st, data);
  this.log_post = () => log_post(this.state, data);   // host evaluation at the current state of chain 0
}

AmwgSampler.prototype._each = function (f) { return this._shards.map(f); };

AmwgSampler.prototype._merge = function (blocks, rows) {   // per-shard [rows][c_shard] -> [rows][chains]
  if (blocks.length === 1) return blocks[0];
  const out = new Float64Array(rows * this.chains);
  this._shards.forEach((sh, k) => {
    for (let r = 0; r < rows; r++) out.set(blocks[k].subarray(r * sh.count, (r + 1) * sh.count), r * this.chains + sh.offset);
  });
  return out;
};

AmwgSampler.prototype.burn = function (n_iterations) {
  const N = native();
  this._each((sh) => N.burnAsync(sh.handle, n_iterations));
  this._each((sh) => N.sync(sh.handle));
};

AmwgSampler.prototype.step = function () { this.burn(1); return this.state; };

AmwgSampler.prototype.sample = function (n_iterations) {
  const N = native(), thin = this.thinning_interval;
  const kept = Math.ceil(n_iterations / thin);
  this._each((sh) => N.sampleAsync(sh.handle, n_iterations, thin));
  // mcmc.js:1009-1013: by default every key of the state is recorded, parameters first, then derived quantities
  const monitored = this.monitored_params === null ? this.param_names.concat(this.derived) : this.monitored_params;
  const C = this.chains, P = this.PR, out = {};
  const derivedLayout = this.derived.map((name, q) => ({ name, base: this.P + q, len: 1, dim: [1], scalar: true }));
  if (C > 1 && this._shards.length === 1) {
    // many chains on one device: every monitored parameter's [kept][len][chains] array is filled straight from the device, launch by
    // launch while the later launches still run (amwg_fetch_draws_slices) -- no [kept][P][chains] intermediate, no second copy in JavaScript
    const found = monitored.map((name) => this._layout.find((l) => l.name === name) || derivedLayout.find((l) => l.name === name));
    const use = found.filter((L) => L);
    const arrays = N.fetchDrawsSplit(this._shards[0].handle, kept, use.map((L) => L.base), use.map((L) => L.len));
    let k = 0;
    monitored.forEach((name, m) => {
      const L = found[m];
      if (!L) { out[name] = []; return; }
      const arr = arrays[k++];
      Object.defineProperty(arr, 'layout', { value: { kept, len: L.len, chains: C, dim: L.dim }, enumerable: false });
      out[name] = arr;
    });
    return out;
  }
  // several shards: each device copies its own block to the host (N PCIe links in parallel: the default), or -- options.gather -- the blocks are
  // first gathered to the device of shard options.gather_root (default 0) by RCCL inside the library and leave in ONE copy (north_star's wording)
  let blocks;
  if (this._shards.length > 1 && this.options && this.options.gather) {
    this._each((sh) => N.sync(sh.handle));
    const g = N.groupGatherDraws(this._shards.map((sh) => sh.handle), this.options.gather_root || 0, kept);
    blocks = this._shards.map((sh, k) => g.draws.subarray(g.offsets[k], g.offsets[k] + kept * this.PR * sh.count));
  } else blocks = this._each((sh) => N.fetchDraws(sh.handle, kept));
  const flat = this._merge(blocks, kept * this.PR);   // [kept][P + derived][chains]
  for (const name of monitored) {
    const L = this._layout.find((l) => l.name === name) || derivedLayout.find((l) => l.name === name);
    if (!L) { out[name] = []; continue; }
    if (C === 1) {                          // reference shape (mcmc.js:1015-1029): one entry per kept draw
      const draws = new Array(kept);
      for (let t = 0; t < kept; t++) draws[t] = L.scalar ? flat[t * P + L.base] : nest(flat, t * P + L.base, L.dim);
      out[name] = draws;
    } else {
      const arr = new Float64Array(kept * L.len * C);
      for (let t = 0; t < kept; t++) arr.set(flat.subarray((t * P + L.base) * C, (t * P + L.base + L.len) * C), t * L.len * C);
      Object.defineProperty(arr, 'layout', { value: { kept, len: L.len, chains: C, dim: L.dim }, enumerable: false });
      out[name] = arr;
    }
  }
  return out;
};

/** Like sample(n), but the draws stay in HBM: nothing is copied to the host; moments(), quantiles() and convergence()
 *  then summarise them on the device (65 536 chains x 5 000 draws are 5 GB -- more than a JS ArrayBuffer holds). */
AmwgSampler.prototype.sample_on_device = function (n_iterations) {
  const N = native();
  this._each((sh) => N.sampleAsync(sh.handle, n_iterations, this.thinning_interval));
  this._each((sh) => N.sync(sh.handle));
  return Math.ceil(n_iterations / this.thinning_interval);
};

Object.defineProperty(AmwgSampler.prototype, 'state', {
  get: function () {
    const N = native(), C = this.chains;
    const flat = this._merge(this._each((sh) => N.getState(sh.handle)), this.P);   // [P][chains]
    const st = {};
    for (const L of this._layout) {
      if (C === 1) st[L.name] = L.scalar ? flat[L.base] : nest(flat, L.base, L.dim);
      else st[L.name] = flat.subarray(L.base * C, (L.base + L.len) * C);
    }
    if (C === 1 && this.derived.length) this._host_log_post(st);   // the closure itself fills in its derived keys (mcmc.js:961-963)
    return st;
  },
});

AmwgSampler.prototype.monitor = function (params_to_monitor) { this.monitored_params = params_to_monitor; };
AmwgSampler.prototype.thin = function (thinning_interval) { this.thinning_interval = thinning_interval; };
AmwgSampler.prototype.start_adaptation = function () { const N = native(); this._each((sh) => N.setAdapting(sh.handle, true)); this._adapting = true; };
AmwgSampler.prototype.stop_adaptation = function () { const N = native(); this._each((sh) => N.setAdapting(sh.handle, false)); this._adapting = false; };

/** Per-parameter stepper state (the content of mcmc.js:563-571), plus run totals.  chains === 1:
 *  plain numbers / nested arrays as in the reference; otherwise typed arrays [element][chain]. */
AmwgSampler.prototype.info = function () {
  const N = native(), C = this.chains;
  const per = this._each((sh) => N.info(sh.handle));
  const keys = ['prop_log_scale', 'acceptance_count', 'iterations_since_adaption', 'batch_count', 'accepts', 'inbounds'];
  const merged = {};
  for (const k of keys) {
    if (per.length === 1) { merged[k] = per[0][k]; continue; }
    const out = new (per[0][k].constructor)(this.P * C);
    this._shards.forEach((sh, s) => { for (let p = 0; p < this.P; p++) out.set(per[s][k].subarray(p * sh.count, (p + 1) * sh.count), p * C + sh.offset); });
    merged[k] = out;
  }
  const steppers = {};
  for (const L of this._layout) {
    const binary = this.params[L.name].type === 'binary';      // BinarySteppers keep no proposal scale or adaptation state (mcmc.js:745-767): run totals only
    const shown = binary ? ['accepts', 'inbounds'] : keys;
    const one = (e) => { const o = {}; for (const k of shown) o[k] = merged[k][(L.base + e) * C]; return o; };
    if (C === 1) steppers[L.name] = L.scalar ? one(0) : nest(Array.from({ length: L.len }, (_, e) => one(e)), 0, L.dim);
    else { const o = {}; for (const k of shown) o[k] = merged[k].subarray(L.base * C, (L.base + L.len) * C); steppers[L.name] = o; }
  }
  return { state: this.state, thin: this.thinning_interval, monitor: this.monitored_params, steppers,
           launch: this._each((sh) => Object.assign({ device: sh.device, chains: sh.count }, N.launchInfo(sh.handle))) };
};

/** Posterior mean / sd per scalar component over all chains x kept draws of the last sample(): device-side reduction; with
 *  `options.devices` every device reduces its shard and the partial sums meet in an RCCL all-reduce (amwg_group_moments). */
AmwgSampler.prototype.moments = function () {
  const N = native();
  const m = this._shards.length === 1 ? N.moments(this._shards[0].handle) : N.groupMoments(this._shards.map((sh) => sh.handle)), out = {};
  for (const L of this._layout) out[L.name] = { mean: Array.from(m.mean.subarray(L.base, L.base + L.len)), sd: Array.from(m.sd.subarray(L.base, L.base + L.len)) };
  this.derived.forEach((name, q) => { out[name] = { mean: [m.mean[this.P + q]], sd: [m.sd[this.P + q]] }; });
  return out;
};

/** Split-R-hat and effective sample size per scalar component (and derived quantity) over the last sample(): device-side
 *  per-chain reduction, chains >= 2; with `options.devices` over the chains of all devices (RCCL all-reduce of the per-device
 *  sums, amwg_group_diagnostics). */
AmwgSampler.prototype.convergence = function () {
  const N = native();
  const d = this._shards.length === 1 ? N.convergence(this._shards[0].handle) : N.groupConvergence(this._shards.map((sh) => sh.handle)), out = {};
  for (const L of this._layout) out[L.name] = { rhat: Array.from(d.rhat.subarray(L.base, L.base + L.len)), ess: Array.from(d.ess.subarray(L.base, L.base + L.len)) };
  this.derived.forEach((name, q) => { out[name] = { rhat: [d.rhat[this.P + q]], ess: [d.ess[this.P + q]] }; });
  return out;
};

/** Posterior quantiles per scalar component (and derived quantity) over all chains x kept draws of the last sample():
 *  device radix sort, R's default (type 7) interpolation.  quantiles([0.025, 0.5, 0.975]) -> {name: [[q...] per element]} */
AmwgSampler.prototype.quantiles = function (probs) {
  const N = native();
  const pr = Float64Array.from(probs), out = {};
  const q = this._shards.length === 1 ? N.quantiles(this._shards[0].handle, pr) : N.groupQuantiles(this._shards.map((sh) => sh.handle), pr);   // several devices: RCCL gather to the first, sorted there
  const row = (c) => Array.from(q.subarray(c * pr.length, (c + 1) * pr.length));
  for (const L of this._layout) out[L.name] = Array.from({ length: L.len }, (_, e) => row(L.base + e));
  this.derived.forEach((name, k) => { out[name] = [row(this.P + k)]; });
  return out;
};

/** Per-chain starting points: f(chainIndex) -> state object shaped like sampler.state of one chain ({name: number | nested array}).
 *  The reference starts from the completed `init` (mcmc.js:954-957); many chains want over-dispersed starts. */
AmwgSampler.prototype.init_chains = function (f) {
  const N = native(), C = this.chains;
  const all = new Float64Array(this.P * C);
  for (let c = 0; c < C; c++) {
    const st = f(c);
    for (const L of this._layout) {
      const vals = flatten(st[L.name], []);
      if (vals.length !== L.len) throw 'init_chains: ' + L.name + ' of chain ' + c + ' does not match dim [' + L.dim + ']';
      for (let e = 0; e < L.len; e++) all[(L.base + e) * C + c] = vals[e];
    }
  }
  this._shards.forEach((sh) => {
    const part = new Float64Array(this.P * sh.count);
    for (let p = 0; p < this.P; p++) part.set(all.subarray(p * C + sh.offset, p * C + sh.offset + sh.count), p * sh.count);
    N.setState(sh.handle, part);
  });
};

AmwgSampler.prototype.diagnostics = function () { const N = native(); return this._each((sh) => N.diag(sh.handle, this.param_names.length)); };
AmwgSampler.prototype.close = function () { const N = native(); this._each((sh) => N.destroy(sh.handle)); this._shards = []; };

// Host-side random helpers the reference module also exports (mcmc.js:31-54, 1104-1106).  They draw from Math.random,
// like the reference's; the sampler itself never calls them (its streams are Philox, on the device).
function runif(min, max) { return Math.random() * (max - min) + min; }
function runif_discrete(min, max) { return Math.floor(Math.random() * (max - min + 1)) + min; }
function rnorm(mean, sd) {      // Leva's ratio-of-uniforms, the same constants as csrc/amwg_kernel.h rnorm_js
  let u, v, x, y, q;
  do {
    u = Math.random();
    v = 1.7156 * (Math.random() - 0.5);
    x = u - 0.449871;
    y = Math.abs(v) + 0.386595;
    q = x * x + y * (0.19600 * y - 0.25472 * x);
  } while (q > 0.27597 && (q > 0.27846 || v * v > -4 * Math.log(u) * u * u));
  return (v / u) * sd + mean;
}

// ---------------------------------------------------------------------------------------------
// Stand-alone steppers (mcmc.js:1109-1115 exports them; tests/test_mcmc_js.R:52-140 drives them directly):
//     new mcmc.RealMetropolisStepper(params, state, log_post, options)   .step() .info() .start_adaptation() .stop_adaptation()
// `params` is a COMPLETE definition (dim, lower, upper -- mcmc.js:417-421) of the parameter(s) the stepper moves, `state` the object
// it shares with the caller (and with other steppers), `log_post` a function of NO arguments that reads that object (mcmc.js:428-431).
// Here every stepper owns a one-chain device sampler whose state holds ALL numeric entries of `state`: the stepper's own parameters
// are stepped, the others are read-only slots (AMWG_FIXED) refreshed from the host object before each step, so several steppers --
// or the caller -- may move different entries of the same object, as with the reference.  log_post is translated like a sampler's
// closure; the free name through which it reaches `state` is recognised by identity (a global, or options.constants).
// One step() is one kernel launch plus two small copies (tens of microseconds): the classes exist for drop-in compatibility;
// steps(n) (not in the reference) runs n steps in one launch.
function stateEntry(v) {
  if (typeof v === 'number' || typeof v === 'boolean') return { dim: [1], flat: [Number(v)], scalar: true };
  if (!Array.isArray(v) || v.length === 0) return null;
  const flat = flatten(v, []);
  if (!flat.every((e) => typeof e === 'number' || typeof e === 'boolean')) return null;
  let dim;
  try { dim = shapeOf(v); } catch (e) { return null; }
  if (dim.reduce((a, b) => a * b, 1) !== flat.length) return null;
  return { dim, flat: flat.map(Number), scalar: false };
}
function writeNested(target, flat, offset) {     // in place, row-major; returns the next offset
  for (let i = 0; i < target.length; i++) {
    if (Array.isArray(target[i])) offset = writeNested(target[i], flat, offset);
    else target[i] = flat[offset++];
  }
  return offset;
}

function Stepper(params, state, log_post) {      // mcmc.js:432-437
  this.params = params;
  this.state = state;
  this.log_post = log_post;
}
Stepper.prototype.step = function () { throw 'Every Stepper need to implement step()'; };
Stepper.prototype.start_adaptation = function () {};
Stepper.prototype.stop_adaptation = function () {};
Stepper.prototype.info = function () { return {}; };

// shared constructor body: `types` maps a parameter name to the type its stepper class implies
function openEngine(self, types, options) {
  options = options || {};
  const names = Object.keys(self.params);
  const completed = {};
  for (const name of names) {
    const p = self.params[name], type = types[name];
    const dim = typeof p.dim === 'number' ? [p.dim] : (p.dim || [1]).slice();
    const cur = stateEntry(self.state[name]);
    if (!cur) throw 'the state has no numeric entry for parameter ' + name;
    if (cur.flat.length !== dim.reduce((a, b) => a * b, 1)) throw 'state.' + name + ' does not match dim [' + dim + ']';
    completed[name] = { type, dim, lower: type === 'binary' ? 0 : (p.lower === undefined ? -Infinity : p.lower),
      upper: type === 'binary' ? 1 : (p.upper === undefined ? Infinity : p.upper), init: isScalarDim(dim) ? cur.flat[0] : nest(cur.flat, 0, dim) };
  }
  let skip = {};
  for (let attempt = 0; ; attempt++) {
    const fixed = [];
    for (const key of Object.keys(self.state)) {
      if (completed.hasOwnProperty(key) || skip[key]) continue;
      const e = stateEntry(self.state[key]);
      if (e) fixed.push({ name: key, dim: e.dim, len: e.flat.length, flat: e.flat, scalar: e.scalar });
    }
    // one lane per chain: the closure's own summation order, so a seeded stepper reproduces the reference bit for bit
    const engineOptions = Object.assign({ lanes_per_chain: 1 }, options, { chains: 1, devices: undefined, [SHARED]: { state: self.state, fixed } });
    try {
      self._engine = new AmwgSampler(completed, self.log_post, undefined, engineOptions);
      self._fixed = fixed;
      break;
    } catch (e) {
      // a derived quantity (`state.key = ...` inside log_post) that an earlier evaluation already left in the state object is not a slot
      const m = /assigns to the parameter state\.(\w+)/.exec(String(e));
      if (!m || completed.hasOwnProperty(m[1]) || skip[m[1]] || attempt > 16) throw e;
      skip[m[1]] = true;
    }
  }
  const E = self._engine;
  self._mirror = Float64Array.from(flatten(names.map((n) => self.state[n]), []).concat(flatten(self._fixed.map((f) => f.flat), [])));
  self._adapting = E._layout.map((L) => flatten(componentOptions(L.name, E.params[L.name], options).map((o) => !!o.is_adapting), []));
}

// host object -> device, if anybody moved an entry since the device last saw it
Stepper.prototype._push = function () {
  const E = this._engine, cur = new Float64Array(E.P);
  let k = 0;
  for (const L of E._layout) for (const v of flatten([this.state[L.name]], [])) cur[k++] = Number(v);
  for (const f of this._fixed) for (const v of flatten([this.state[f.name]], [])) cur[k++] = Number(v);
  if (k !== E.P) throw 'the shape of the state object changed since the stepper was created';
  let same = true;
  for (let i = 0; i < k && same; i++) same = Object.is(cur[i], this._mirror[i]);
  if (!same) { native().setState(E._shards[0].handle, cur); this._mirror = cur; }
};
// device -> host object (only the entries this stepper owns can have moved)
Stepper.prototype._pull = function () {
  const E = this._engine, flat = native().getState(E._shards[0].handle);
  for (const L of E._layout) {
    if (Array.isArray(this.state[L.name])) writeNested(this.state[L.name], flat, L.base);
    else this.state[L.name] = flat[L.base];
  }
  this._mirror = Float64Array.from(flat);
  if (E.derived.length) this.log_post();    // the closure itself refreshes its derived keys at the new state
};
Stepper.prototype._value = function (name) {
  const L = this._engine._layout.find((l) => l.name === name), flat = this._mirror;
  return Array.isArray(this.state[name]) ? nest(flat, L.base, L.dim) : flat[L.base];
};
/** n steps in one launch (not in the reference); returns what step() returns after the last one. */
Stepper.prototype.steps = function (n) {
  this._push();
  this._engine.burn(n);
  this._pull();
  return this._result();
};
Stepper.prototype.close = function () { if (this._engine) this._engine.close(); };

function deviceStepper(className, type, shape) {
  const oneParam = { scalar: className + ' can only handle one parameter.', multi: className + " can't handle more than one parameter." };
  const C = function (params, state, log_post, options) {
    Stepper.call(this, params, state, log_post);
    const names = Object.keys(this.params);
    if (names.length !== 1) throw (shape === 'scalar' && type !== 'binary') ? oneParam.scalar : oneParam.multi;    // mcmc.js:489, 634, 746, 788
    this.param_name = names[0];
    const dim = this.params[this.param_name].dim;
    if (shape === 'scalar' && type !== 'binary' && !(Array.isArray(dim) && isScalarDim(dim)))
      throw className + ' can only handle one one-dimensional parameter.';                                      // mcmc.js:494
    openEngine(this, { [this.param_name]: type }, options);
  };
  C.prototype = Object.create(Stepper.prototype);
  C.prototype.constructor = C;
  C.prototype._result = function () { return this._value(this.param_name); };
  C.prototype.step = function () { return this.steps(1); };
  if (type !== 'binary') {
    C.prototype.start_adaptation = function () { this._engine.start_adaptation(); this._adapting = this._adapting.map((a) => a.map(() => true)); };
    C.prototype.stop_adaptation = function () { this._engine.stop_adaptation(); this._adapting = this._adapting.map((a) => a.map(() => false)); };
    C.prototype.info = function () { return stepperInfo(this, this._engine._layout[0], 0); };
  }
  return C;
}

// the content of OnedimMetropolisStepper.info (mcmc.js:563-571), per component, nested like the parameter (mcmc.js:700-704)
function stepperInfo(self, L, li) {
  const raw = native().info(self._engine._shards[0].handle);
  const one = (e) => ({ prop_log_scale: raw.prop_log_scale[L.base + e], is_adapting: self._adapting[li][e], acceptance_count: raw.acceptance_count[L.base + e],
    iterations_since_adaption: raw.iterations_since_adaption[L.base + e], batch_count: raw.batch_count[L.base + e] });
  return L.scalar ? one(0) : nest(Array.from({ length: L.len }, (_, e) => one(e)), 0, L.dim);
}

const RealMetropolisStepper = deviceStepper('OnedimMetropolisStepper', 'real', 'scalar');
const IntMetropolisStepper = deviceStepper('OnedimMetropolisStepper', 'int', 'scalar');
const MultiRealComponentMetropolisStepper = deviceStepper('MultidimComponentMetropolisStepper', 'real', 'multi');
const MultiIntComponentMetropolisStepper = deviceStepper('MultidimComponentMetropolisStepper', 'int', 'multi');
const BinaryStepper = deviceStepper('BinaryStepper', 'binary', 'scalar');
const BinaryComponentStepper = deviceStepper('BinaryComponentStepper', 'binary', 'multi');

/** mcmc.js:837-916: one sub-stepper per parameter, picked by type and dim, visited in a shuffled order that persists. */
function AmwgStepper(params, state, log_post, options) {
  Stepper.call(this, params, state, log_post);
  this.param_names = Object.keys(this.params);
  const types = {};
  for (const name of this.param_names) {
    const t = params[name].type;
    if (t !== 'real' && t !== 'int' && t !== 'binary') throw "AmwgStepper can't handle parameter " + name + ' with type ' + t;   // mcmc.js:867
    types[name] = t;
  }
  openEngine(this, types, options);
}
AmwgStepper.prototype = Object.create(Stepper.prototype);
AmwgStepper.prototype.constructor = AmwgStepper;
AmwgStepper.prototype._result = function () { return this.state; };
AmwgStepper.prototype.step = function () { return this.steps(1); };
AmwgStepper.prototype.start_adaptation = function () { this._engine.start_adaptation(); this._adapting = this._adapting.map((a) => a.map(() => true)); };
AmwgStepper.prototype.stop_adaptation = function () { this._engine.stop_adaptation(); this._adapting = this._adapting.map((a) => a.map(() => false)); };
AmwgStepper.prototype.info = function () {     // keyed by the parameter each entry really belongs to (the reference's labels go stale
  const out = {};                              // after the first shuffle, mcmc.js:887 vs :909)
  this._engine._layout.forEach((L, li) => { out[L.name] = this._engine.params[L.name].type === 'binary' ? (L.scalar ? {} : nest(Array.from({ length: L.len }, () => ({})), 0, L.dim)) : stepperInfo(this, L, li); });
  return out;
};

/** {hits, misses, dir} of the on-disk cache of compiled closures in this process ($AMWG_CACHE_DIR | $XDG_CACHE_HOME/amwg | ~/.cache/amwg). */
function code_cache_stats() { return native().codeCacheStats(); }

module.exports = { code_cache_stats, runif, runif_discrete, rnorm, AmwgSampler, complete_params, param_init_fixed, componentOptions, models, ld, native, translate: translator.translate,
  RealMetropolisStepper, IntMetropolisStepper, MultiRealComponentMetropolisStepper, MultiIntComponentMetropolisStepper, BinaryStepper, BinaryComponentStepper, AmwgStepper };
