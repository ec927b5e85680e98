'use strict';
/*
 * translate.js -- turns a user's `log_post(state, data)` closure (mcmc.js:958-960; README.md:18-43,
 * 149-164; tests/test_data.js:80-211) into the HIP source of `struct amwg::UserModel`, which
 * libamwg.so compiles with hiprtc together with the fused step kernel (csrc/amwg_kernel.h).
 *
 * The closure is read from its own source text (Function.prototype.toString) and must stay
 * inside a numeric subset of JavaScript:
 *     var/let/const, assignments (= += -= *= /= %=, ++ --), for / while / do-while / if / else / switch (no fall-through) / break / continue / return, blocks
 *     numbers, + - * / % **, | & ^ ~ << >> >>> (ToInt32 semantics), comparisons, && || !, ?:, Math.{log,exp,log1p,expm1,log10,log2,pow,sqrt,cbrt,hypot,abs,floor,ceil,round,trunc,sign,
 *     min,max,sin,cos,tan,asin,acos,atan,atan2,sinh,cosh,tanh,asinh,acosh,atanh,imul,clz32,fround,PI,E,...}, isNaN, isFinite, every ld.* of distributions.js (array-valued ones unrolled),
 *     state.name, state.name[i][j], data.field, data.field[i][j], data.field.length, arrays of records (data[i].x, var row = data[i]), categorical strings in the data (only compared: row.group === 'control', or looked up: LEVELS.indexOf(row.group)), local
 *     aliases of those (var p = par.p[0]), derived quantities (state.key = expr, mcmc.js:961-963),
 *     helper functions and constants passed in options.helpers / options.constants (or globals); helpers that take numbers become
 *     device functions, helpers that are handed the state, the data or arrays of them (`log_prior(state) + log_lik(state, data)`) are
 *     inlined (they need a single return at their end);
 *     post-ES5 spellings, rewritten into the above while parsing: destructuring in parameters and declarations
 *     (`({mu, sigma}, {x}) => ...`, `const [a, , b] = state.theta`), `for (const x of arr)`, `arr.forEach(cb)` as a statement
 *     (`return` inside the callback = continue), `arr.reduce(cb[, init])`, `arr.some(cb)`, `arr.every(cb)` anywhere in the expressions of a statement
 *     (callback parameters and locals are renamed apart; the reduce becomes its own sequential accumulator, as in JS),
 *     `arr.map(cb)`, `Array(n)`, `new Array(n)`, `Array(n).fill(v)` as local arrays of a length known when the sampler is built (<= 2048);
 *     `var a = []` grown by one `a.push(v)` per iteration of a counted loop from 0 (the same thing); let/const block scoping.
 * Anything else throws a string that says what is not supported (no CPU fallback).  A read outside an array, or with a non-integer
 * index, yields NaN as in JavaScript (undefined in arithmetic) and touches no memory; indices that provably stay inside (loop counters,
 * elements of integer data arrays, + - * % of those) are not checked at run time.
 *
 * Fidelity.  Every JavaScript number operation becomes the same IEEE fp64 operation in the same
 * order (the kernel is compiled with -ffp-contract=off); Math.exp/log are the bit-identical V8
 * twins of csrc/amwg_math.h; constant sub-expressions are folded HERE, by V8 itself.  With one lane
 * per chain the generated body is the closure's own evaluation order, so a seeded run reproduces
 * the reference bit for bit.  With G lanes per chain the top-level loops that only accumulate into
 * the returned variable -- or into any of several running sums that are only ever combined linearly
 * (linearAccumulators) -- are split G ways (lane j takes iterations j, j+G, ...; lane 0 also adds
 * every term outside those loops) and the kernel adds the lane partial sums with an xor butterfly.
 *
 * Result-preserving optimisations, applied only where a value provably does not change inside a loop: ld.norm with a
 * loop-invariant sd uses the hoisted form of csrc/amwg_user.h (same roundings, correctly rounded quotient; a straight-line
 * fast loop plus an IEEE replay if a range precondition failed); ld.bern with a loop-invariant p selects between its two
 * possible values and, over a whole 0/1 array with one lane per chain, fast-forwards the two-valued sequential sum exactly
 * (csrc/amwg_twoval.h); pure calls with loop-invariant arguments are evaluated once before the loop; lfactorial / lchoose of
 * pure data are tabulated on the host by the same formula; integer-valued data arrays are stored as u8 / i32.
 */

// ------------------------------------------------------------------------------------------
// exact C++ literals
function hexFloat(v) {
  if (Number.isNaN(v)) return '__builtin_nan("")';
  if (v === Infinity) return 'kInf';
  if (v === -Infinity) return '(-kInf)';
  if (Number.isInteger(v) && Math.abs(v) < 9007199254740992 && !(v === 0 && 1 / v < 0)) return v.toFixed(1);
  const buf = new DataView(new ArrayBuffer(8));
  buf.setFloat64(0, v);
  const hi = buf.getUint32(0), lo = buf.getUint32(4);
  const sign = hi >>> 31 ? '-' : '';
  const exp = (hi >>> 20) & 0x7ff;
  const mant = ((hi & 0xfffff).toString(16).padStart(5, '0') + lo.toString(16).padStart(8, '0'));
  if (exp === 0) {
    if ((hi & 0xfffff) === 0 && lo === 0) return sign + '0.0';
    return sign + '0x0.' + mant + 'p-1022';
  }
  return sign + '0x1.' + mant + 'p' + (exp - 1023 >= 0 ? '+' : '') + (exp - 1023);
}

// norm_inv(<literal>) folded at translation time (round 6): the loop invariants of ld.norm with a CONSTANT sd -- c = -0.5 log(2 pi) - log(sd), den = 2 sd sd, the
// double-double reciprocal of csrc/amwg_div.h -- as literals, instead of a logarithm and a division every time the generated code passes the statement (a prior
// `lp += ld.norm(theta[k], mu, 10)` inside the head of a row plan is evaluated several times per step).  Math.log is V8's -- the function log_v8 restates bit for bit
// (tests/test_oracle_math.py) --, the products and the quotient are IEEE, and the reciprocal's low word RN(fma(-den, hi, 1) * hi) is formed from the EXACT residual
// (BigInt arithmetic on the significands; it is representable, Markstein) and one rounded product: the same bits as norm_inv() on the device, which the host build
// of the generated text checks (tests/host/user_eval_host.cpp evaluates both forms).
function parseLiteral(t) {
  t = t.trim();
  if (/^-?\d+(?:\.\d*)?(?:e[+-]?\d+)?$/i.test(t)) return Number(t);
  const m = /^(-?)0x([01])\.([0-9a-f]{13})p([+-]?\d+)$/i.exec(t);
  if (!m) return NaN;
  const mant = (BigInt(m[2]) << 52n) | BigInt('0x' + m[3]);
  return (m[1] ? -1 : 1) * Number(mant) * Math.pow(2, Number(m[4]) - 52);
}
function decompose(v) {      // v = m * 2^e exactly, m a BigInt (v finite, positive)
  const buf = new DataView(new ArrayBuffer(8));
  buf.setFloat64(0, v);
  const hi = buf.getUint32(0), lo = buf.getUint32(4), ex = (hi >>> 20) & 0x7ff;
  const frac = (BigInt(hi & 0xfffff) << 32n) | BigInt(lo);
  return ex === 0 ? { m: frac, e: -1074 } : { m: frac | (1n << 52n), e: ex - 1075 };
}
function foldNormInv(sd) {
  if (!(sd > 0) || !isFinite(sd)) return null;
  const c = (-0.5 * Math.log(2 * Math.PI)) - Math.log(sd);
  const den = (2 * sd) * sd;
  if (!(den >= Math.pow(2, -200) && den <= Math.pow(2, 200))) return null;      // (mid_range: outside it the device takes IEEE division; leave the call)
  const hi = 1 / den;
  const a = decompose(den), b = decompose(hi);
  const E = a.e + b.e;      // den * hi = a.m b.m 2^E, next to 1: E is about -104
  if (E >= 0) return null;
  const r = (1n << BigInt(-E)) - a.m * b.m;      // (1 - den * hi) * 2^-E, exact
  let rn = r < 0n ? -r : r, sh = 0;
  while (rn !== 0n && (rn & 1n) === 0n) { rn >>= 1n; sh++; }
  if (rn >= (1n << 53n)) return null;            // (cannot happen for a correctly rounded reciprocal: the residual is representable)
  const resid = (r < 0n ? -1 : 1) * Number(rn) * Math.pow(2, E + sh);
  const lo = resid * hi;
  return 'NormInv{' + hexFloat(c) + ', ' + hexFloat(den) + ', Reciprocal{' + hexFloat(hi) + ', ' + hexFloat(lo) + '}, true}';
}
// ... and the scalar densities with literal parameters, evaluated by the head of a closure on one lane several times per step: ld_norm(x, mean, <sd literal>) -> the
// same expression tree with its two constants folded (c = -0.5 log(2 pi) - log(sd), den = (2 sd) sd: csrc/amwg_ld.h ld_norm_c); ld_unif(x, <lo>, <hi>) -> its constant
// log(1 / (hi - lo)) folded (ld_unif_c).  Same operations on the same values: the logarithms are V8's either way.
function splitCallArgs(text, open) {      // text[open] === '(' -> {args: [...], end: index after ')'} or null
  let depth = 0, start = open + 1;
  const args = [];
  for (let i = open; i < text.length; i++) {
    const ch = text[i];
    if (ch === '(') depth++;
    else if (ch === ')') { depth--; if (depth === 0) { args.push(text.slice(start, i).trim()); return { args, end: i + 1 }; } }
    else if (ch === ',' && depth === 1) { args.push(text.slice(start, i).trim()); start = i + 1; }
    else if (ch === '\n' || ch === ';') return null;
  }
  return null;
}
function foldConstantDensities(text) {
  let out = '', at = 0;
  const re = /\b(ld_norm|ld_unif)\(/g;
  for (let m; (m = re.exec(text));) {
    if (m.index < at) continue;
    const call = splitCallArgs(text, m.index + m[1].length);
    if (!call || call.args.length !== 3) continue;
    let repl = null;
    if (m[1] === 'ld_norm') {
      const sd = parseLiteral(call.args[2]);
      if (sd > 0 && isFinite(sd)) repl = 'ld_norm_c(' + call.args[0] + ', ' + call.args[1] + ', ' + hexFloat((-0.5 * Math.log(2 * Math.PI)) - Math.log(sd)) + ', ' + hexFloat((2 * sd) * sd) + ') /* ld_norm(., ., ' + call.args[2] + ') */';
    } else {
      const lo = parseLiteral(call.args[1]), hi = parseLiteral(call.args[2]);
      if (!Number.isNaN(lo) && !Number.isNaN(hi)) repl = 'ld_unif_c(' + call.args[0] + ', ' + call.args[1] + ', ' + call.args[2] + ', ' + hexFloat(Math.log(1 / (hi - lo))) + ')';
    }
    if (repl) { out += text.slice(at, m.index) + repl; at = call.end; re.lastIndex = call.end; }
  }
  return out + text.slice(at);
}
function foldConstantNormInv(text) {
  text = foldConstantDensities(text);
  return text.replace(/norm_inv\((-?(?:0x[0-9a-fp.+-]+|[0-9][0-9.e+-]*))\)/gi, (whole, lit) => {
    const v = parseLiteral(lit);
    const f = Number.isNaN(v) ? null : foldNormInv(v);
    return f ? f + ' /* norm_inv(' + lit + ') */' : whole;
  });
}

const { tokenize, parseFunctionSource, desugarBlock, walk, assignedNames, definitelyAssigned, idsOf, containsKind, declaredIn } = require('./parse.js');


// the densities of distributions.js with scalar arguments: name -> [device function, arity]
const LD_FUNS = {
  norm: ['ld_norm', 3], unif: ['ld_unif', 3], beta: ['ld_beta', 3], bern: ['ld_bern', 2], pois: ['ld_pois', 2],
  cauchy: ['ld_cauchy', 3], laplace: ['ld_laplace', 3], dexp: ['ld_laplace', 3], gamma: ['ld_gamma', 3],
  invgamma: ['ld_invgamma', 3], lnorm: ['ld_lnorm', 3], pareto: ['ld_pareto', 3], t: ['ld_t', 4], weibull: ['ld_weibull', 3],
  logis: ['ld_logis', 3], exp: ['ld_exp', 2], binom: ['ld_binom', 3], nbinom: ['ld_nbinom', 3], hyper: ['ld_hyper', 4],
  lgamma: ['lgamma_js', 1], lfactorial: ['lfactorial_js', 1], lchoose: ['lchoose_js', 2], lbeta: ['lbeta_js', 2],
};
const MATH_CONST = { PI: Math.PI, E: Math.E, LN2: Math.LN2, LN10: Math.LN10, LOG2E: Math.LOG2E, LOG10E: Math.LOG10E, SQRT2: Math.SQRT2, SQRT1_2: Math.SQRT1_2 };
// Math.f -> [device function, arity, host function used for constant folding]
const MATH_FUNS = {
  log: ['log_v8', 1, Math.log], exp: ['exp_v8', 1, Math.exp], log1p: ['log1p_v8', 1, Math.log1p], expm1: ['expm1_v8', 1, Math.expm1],
  tanh: ['tanh_v8', 1, Math.tanh], atan: ['atan_v8', 1, Math.atan], log10: ['log10_v8', 1, Math.log10], sqrt: ['__builtin_sqrt', 1, Math.sqrt], abs: ['__builtin_fabs', 1, Math.abs],
  floor: ['__builtin_floor', 1, Math.floor], ceil: ['__builtin_ceil', 1, Math.ceil], round: ['js_round', 1, Math.round],
  trunc: ['js_trunc', 1, Math.trunc], sign: ['js_sign', 1, Math.sign],
  sin: ['sin_v8', 1, Math.sin], cos: ['cos_v8', 1, Math.cos], tan: ['tan_v8', 1, Math.tan], asin: ['asin_v8', 1, Math.asin], acos: ['acos_v8', 1, Math.acos],
  sinh: ['sinh_v8', 1, Math.sinh], cosh: ['cosh_v8', 1, Math.cosh], asinh: ['asinh_v8', 1, Math.asinh], acosh: ['acosh_v8', 1, Math.acosh], atanh: ['atanh_v8', 1, Math.atanh],
  cbrt: ['cbrt_v8', 1, Math.cbrt], log2: ['log2_v8', 1, Math.log2], clz32: ['js_clz32', 1, Math.clz32], fround: ['js_fround', 1, Math.fround],
};
const HEAVY = new Set(['log_v8', 'exp_v8', 'pow_v8', 'log1p_v8', 'log1p_exp_v8', 'expm1_v8', 'tanh_v8', 'atan_v8', 'log10_v8', 'sin_v8', 'cos_v8', 'tan_v8', 'asin_v8', 'acos_v8', 'sinh_v8', 'cosh_v8', 'asinh_v8', 'acosh_v8', 'atanh_v8', 'cbrt_v8', 'log2_v8', 'atan2_v8', 'hypot', 'ld_norm', 'ld_beta', 'ld_pois', 'ld_bern', 'ld_gamma', 'ld_invgamma', 'ld_lnorm', 'ld_t',
  'ld_weibull', 'ld_logis', 'ld_binom', 'ld_nbinom', 'ld_hyper', 'ld_cauchy', 'ld_pareto', 'ld_exp', 'ld_laplace', 'ld_unif', 'lgamma_js', 'lfactorial_js', 'lchoose_js', 'lbeta_js']);

const ld_host = require('./ld.js');

// ld.norm calls with a hoisted sd are emitted as NORMCALL(x, m, k RANGEARGS); the finished loop decides the form:
//   'inv'   per-term range check (loops that are not lane-split)
//   'fast'  4-operation quotient + range recording      'slow'  IEEE division
function renderNorm(lines, mode) {
  const call = { inv: 'ld_norm_inv', fast: 'ld_norm_fast', slow: 'ld_norm_slow' }[mode];
  return lines.map((ln) => ln.split('NORMCALL').join(call).split(' RANGEARGS').join(mode === 'fast' ? ', rlo_, rhi_' : '')
    .replace(/ KFASTCHECK\((\w+)\)/g, mode === 'fast' ? ' if (!$1.fast) rlo_ = 0u;' : ''));
}
const hasNormCall = (lines) => lines.some((ln) => ln.indexOf('NORMCALL') >= 0);

// ------------------------------------------------------------------------------------------
function Translator(fn, params, data, opts, isHelper) {
  this.opts = opts || {};
  this.isHelper = !!isHelper;
  if (fn && fn.k === 'Func') this.ast = { params: fn.params, body: fn.body };     // a function expression inside log_post
  else {
    this.fnSource = typeof fn === 'string' ? fn : Function.prototype.toString.call(fn);
    this.ast = parseFunctionSource(this.fnSource);
  }
  this.data = data;
  if (!isHelper && this.opts.state_object) {
    // a stepper's log_post (mcmc.js:424-431: called with NO arguments, it reads the state object it closes over): the state is whatever
    // free name holds opts.state_object; `function () { return dens(state); }` just hands it on, so `dens` is what gets translated
    if (this.ast.params.length > 0) throw 'the log_post of a stepper takes no arguments (it reads the state object it closes over)';
    this.stateName = null;
    this.dataName = null;
    this.resolveForwarding();
  } else {
    if (!isHelper && (this.ast.params.length < 1 || this.ast.params.length > 2)) throw 'log_post must take (state) or (state, data)';
    this.stateName = isHelper ? null : this.ast.params[0];
    this.dataName = isHelper ? null : (this.ast.params[1] || null);
  }
  // parameter layout in Object.keys order (mcmc.js:839), flattened row-major
  this.layout = {};
  let base = 0;
  for (const name of Object.keys(params)) {
    const dim = params[name].dim.slice();
    const len = dim.reduce((a, b) => a * b, 1);
    this.layout[name] = { base, dim, len, scalar: dim.length === 1 && dim[0] === 1 };
    base += len;
  }
  this.P = base;
  // binary parameters need the BinaryStepper branch of the step kernel (two more inlined log_post evaluations per slot); a stand-alone
  // stepper's parameter list may not carry the types of everything it steps, so it always keeps the branch
  this.hasBinary = !!this.opts.state_object || Object.keys(params).some((n) => params[n].type === 'binary');
  this.arrays = [];          // {key, flat: Float64Array, dims}
  this.arrayIds = new Map();
  this.derived = [];         // names, in order of first assignment
  this.localTypes = {};      // name -> 'int' | 'double'
  this.aliases = {};         // name -> symbolic value
  this.helpers = {};         // name -> {cname, nargs}
  this.helperSources = [];
  this.tmp = 0;
  this.heavy = false;
}

Translator.prototype.fail = function (msg) { throw 'AmwgSampler (MI355X): cannot translate log_post: ' + msg; };

// the value a free name of the closure has, as far as the translator can see it: options.constants / options.helpers, then globals
Translator.prototype.freeValue = function (name) {
  const consts = this.opts.constants || {}, helpers = this.opts.helpers || {};
  if (Object.prototype.hasOwnProperty.call(consts, name)) return consts[name];
  if (Object.prototype.hasOwnProperty.call(helpers, name)) return helpers[name];
  if (typeof globalThis !== 'undefined' && name in globalThis) return globalThis[name];
  return undefined;
};
Translator.prototype.isStateName = function (name) {
  if (this.stateName !== null && this.stateName !== undefined) return name === this.stateName;
  return !!this.opts.state_object && this.freeValue(name) === this.opts.state_object;
};

// 'number' | 'object' | null for a path rooted at the state or the data, from the parameter layout and the data itself (no code is emitted)
Translator.prototype.staticKind = function (e) {
  const chain = [];
  let root = e;
  while (root && (root.k === 'Member' || root.k === 'Index')) { chain.unshift(root); root = root.obj; }
  if (!root || root.k !== 'Id') return null;
  if (this.isStateName(root.name)) {
    if (!chain.length) return 'object';
    const first = chain[0];
    if (first.k !== 'Member' || !Object.prototype.hasOwnProperty.call(this.layout, first.prop)) return null;
    const L = this.layout[first.prop], depth = chain.length - 1;
    if (L.scalar) return depth === 0 ? 'number' : null;
    return depth >= L.dim.length ? 'number' : 'object';
  }
  if (this.dataName && root.name === this.dataName) {
    let v = this.data;
    for (const step of chain) {
      if (v === null || v === undefined) return null;
      if (step.k === 'Member') { if (step.prop === 'length') return 'number'; v = v[step.prop]; }
      else v = (Array.isArray(v) || ArrayBuffer.isView(v)) ? v[0] : undefined;
    }
    if (typeof v === 'number' || typeof v === 'boolean') return 'number';
    return (v && typeof v === 'object') ? 'object' : null;
  }
  return null;
};

// functions of the surrounding program that are handed the state / the data (`return log_prior(s) + log_lik(s, d)`) are inlined
Translator.prototype.inlineObjectCalls = function () {
  const own = declaredIn(this.ast.body, new Set(this.ast.params));
  const cache = {};
  const P = { uniq: 1000, fresh(stem) { this.uniq++; return '__' + stem + this.uniq; } };
  P.env = {
    isObject: (a) => this.staticKind(a) === 'object',
    funcOf: (name) => {
      if (own.has(name)) return null;
      if (!Object.prototype.hasOwnProperty.call(cache, name)) {
        const f = this.freeValue(name);
        cache[name] = typeof f === 'function' ? parseFunctionSource(Function.prototype.toString.call(f)) : null;
      }
      return cache[name];
    },
  };
  let needed = false;
  walk(this.ast.body, (x) => { if (x.k === 'Call' && x.callee.k === 'Id' && x.args.some((a) => P.env.isObject(a)) && P.env.funcOf(x.callee.name)) needed = true; });
  if (!needed) return;
  try { this.ast = { params: this.ast.params, body: desugarBlock(this.ast.body, P) }; }
  catch (e) { if (typeof e === 'string' && e.indexOf('AmwgSampler') !== 0) this.fail(e); throw e; }
};

// `function () { return f(state); }` (and chains of such): continue with f, its parameter bound to the state
Translator.prototype.resolveForwarding = function () {
  for (let depth = 0; depth < 8; depth++) {
    const body = this.ast.body.body;
    if (body.length !== 1 || body[0].k !== 'Return' || !body[0].arg || body[0].arg.k !== 'Call') return;
    const call = body[0].arg;
    if (call.callee.k !== 'Id' || !call.args.every((a) => a.k === 'Id')) return;
    const f = this.freeValue(call.callee.name);
    if (typeof f !== 'function') return;
    const statePos = call.args.findIndex((a) => this.isStateName(a.name));
    if (statePos < 0 || call.args.length > 2) return;
    const dataPos = call.args.length === 2 ? 1 - statePos : -1;
    if (dataPos >= 0) {
      const dv = call.args[dataPos].name === this.dataName ? this.data : this.freeValue(call.args[dataPos].name);
      if (dv === undefined) return;
      this.data = dv;
    }
    this.fnSource = Function.prototype.toString.call(f);
    this.ast = parseFunctionSource(this.fnSource);
    if (this.ast.params.length <= statePos) throw 'log_post forwards the state to ' + call.callee.name + '(), which takes fewer arguments';
    this.stateName = this.ast.params[statePos];
    this.dataName = dataPos >= 0 ? (this.ast.params[dataPos] || null) : null;
  }
};

// ---- data arrays -----------------------------------------------------------------------------
function shapeOfData(v) {
  if (ArrayBuffer.isView(v)) return [v.length];
  if (!Array.isArray(v)) return null;
  if (v.length === 0) return [0];
  if (Array.isArray(v[0]) || ArrayBuffer.isView(v[0])) {
    const inner = shapeOfData(v[0]);
    if (!inner) return null;
    for (const e of v) { const s = shapeOfData(e); if (!s || s.join() !== inner.join()) return null; }
    return [v.length].concat(inner);
  }
  for (const e of v) if (typeof e !== 'number' && typeof e !== 'boolean') return null;
  return [v.length];
}
function flattenData(v, out) { if (Array.isArray(v) || ArrayBuffer.isView(v)) { for (let i = 0; i < v.length; i++) flattenData(v[i], out); } else out.push(Number(v)); return out; }

Translator.prototype.registerArray = function (key, value) {
  if (this.arrayIds.has(key)) return this.arrayIds.get(key);
  const dims = shapeOfData(value);
  if (!dims) this.fail('data' + key + ' is not a rectangular array of numbers');
  const flat = Float64Array.from(flattenData(value, []));
  const id = this.arrays.length;
  // device storage: small non-negative integers as u8, other 32-bit integers as i32 (exact), everything else f64
  let u8 = flat.length > 0, i32 = flat.length > 0, is01 = flat.length > 0;
  for (let i = 0; i < flat.length; i++) {
    const v = flat[i];
    if (!(Number.isInteger(v) && !(v === 0 && 1 / v < 0))) { u8 = i32 = is01 = false; break; }
    if (v < 0 || v > 255) u8 = false;
    if (v < -2147483648 || v > 2147483647) i32 = false;
    if (v !== 0 && v !== 1) is01 = false;
  }
  const type = this.opts.f64_arrays ? 0 : (u8 ? 1 : (i32 ? 2 : 0));
  this.arrays.push({ key, flat, dims, type, is01: is01 && type === 1, ctype: ['double', 'uint8_t', 'int32_t'][type], esize: [8, 1, 4][type] });
  this.arrayIds.set(key, id);
  return id;
};

// ---- symbolic values -----------------------------------------------------------------------------
// An int-typed value (loop variables, integer constants, integer-typed data elements and + - * of those) keeps two
// spellings: `code` in 32-bit int arithmetic, used only for array indices, and `dcode`, the same expression in double
// arithmetic, used wherever the value is a JavaScript number (so nothing can overflow where JS would not).
const num = (code, int, cst, dcode) => ({ t: 'num', code, int: !!int, cst, dcode: int ? (dcode || '(double)(' + code + ')') : undefined });
const cnum = (v) => {
  const isInt = Number.isInteger(v) && Math.abs(v) < 2147483648 && !(v === 0 && 1 / v < 0);
  return { t: 'num', code: isInt ? String(v) : hexFloat(v), int: isInt, cst: v };
};
// integer interval a value is known to lie in at translation time (loop counters of constant bounds, elements of integer data arrays,
// + - * % of those); null = unknown.  Array reads whose index provably stays inside need no run-time check.
const rangeOf = (v) => (v && v.t === 'num' ? (v.cst !== undefined ? (Number.isInteger(v.cst) ? [v.cst, v.cst] : null) : (v.int && v.range ? v.range : null)) : null);
const withRange = (v, r) => { if (r && Number.isFinite(r[0]) && Number.isFinite(r[1]) && Math.abs(r[0]) < 2147483648 && Math.abs(r[1]) < 2147483648) v.range = r; return v; };
Translator.prototype.asD = function (v) {
  if (v.t === 'num') {
    if (v.cst !== undefined) return hexFloat(v.cst);
    return v.int ? v.dcode : v.code;
  }
  if (v.t === 'bool') return '((' + v.code + ') ? 1.0 : 0.0)';
  this.fail('a ' + this.describe(v) + ' is used where a number is needed');
};
Translator.prototype.asI = function (v) {
  if (v.t === 'num') return v.int ? v.code : '(int)(' + v.code + ')';
  this.fail('a ' + this.describe(v) + ' is used as an array index');
};
Translator.prototype.asB = function (v) {
  if (v.t === 'bool') return v.code;
  if (v.t === 'num') { const d = this.asD(v); return '((' + d + ') != 0.0 && (' + d + ') == (' + d + '))'; }
  this.fail('a ' + this.describe(v) + ' is used as a condition');
};
Translator.prototype.describe = function (v) {
  return { localArr: 'local array', recArr: 'array of records', rec: 'record', strlit: 'string', strv: 'string', strArr: 'array of strings', strList: 'array of strings', stateObj: 'state object', stateArr: 'parameter array', dataObj: 'data object', dataArr: 'data array', ns: 'namespace', fn: 'function', bool: 'boolean', num: 'number' }[v.t] || v.t;
};

// ---- expressions -----------------------------------------------------------------------------------
Translator.prototype.lookup = function (name) {
  if (name === this.stateName) return { t: 'stateObj' };
  if (this.stateName === null && !this.isHelper && this.opts.state_object && !Object.prototype.hasOwnProperty.call(this.aliases, name) &&
      !Object.prototype.hasOwnProperty.call(this.localTypes, name) && this.freeValue(name) === this.opts.state_object) return { t: 'stateObj' };
  if (this.dataName && name === this.dataName) return this.dataValue('', this.data);
  if (Object.prototype.hasOwnProperty.call(this.aliases, name)) return this.aliases[name];
  if (Object.prototype.hasOwnProperty.call(this.localTypes, name))
    return withRange(num('v_' + name, this.localTypes[name] === 'int', undefined, '(double)v_' + name), this.counterRange && this.counterRange[name]);
  if (name === 'ld') return { t: 'ns', name: 'ld' };
  if (name === 'Math') return { t: 'ns', name: 'Math' };
  if (name === 'isNaN' || name === 'isFinite') return { t: 'fn', ns: 'global', name };
  if (name === 'Number') return { t: 'ns', name: 'Number' };
  if (name === 'Infinity') return cnum(Infinity);
  if (name === 'NaN') return cnum(NaN);
  const consts = this.opts.constants || {};
  if (Object.prototype.hasOwnProperty.call(consts, name)) return this.dataValue('#' + name, consts[name]);
  const helpers = this.opts.helpers || {};
  if (this.localFuncs && Object.prototype.hasOwnProperty.call(this.localFuncs, name)) return { t: 'fn', ns: 'helper', name };
  if (Object.prototype.hasOwnProperty.call(helpers, name)) return { t: 'fn', ns: 'helper', name };
  // script-style code (the reference's README and tests run in a browser page): free names that are GLOBALS are visible -- numbers and
  // arrays are captured by value now, functions become helpers; module-scoped variables are not reachable, they need options.*
  if (typeof globalThis !== 'undefined' && name in globalThis && ['ld', 'Math'].indexOf(name) < 0) {
    const gv = globalThis[name];
    if (typeof gv === 'function') { this.opts.helpers = Object.assign({}, this.opts.helpers, { [name]: gv }); return { t: 'fn', ns: 'helper', name }; }
    if (typeof gv === 'number' || typeof gv === 'boolean' || Array.isArray(gv) || ArrayBuffer.isView(gv) || (gv && typeof gv === 'object'))
      return this.dataValue('#global:' + name, gv);
  }
  // name every free variable of the closure at once (the user fixes them in one go)
  const known = new Set(['ld', 'Math', 'Number', 'isNaN', 'isFinite', 'Infinity', 'NaN', 'Array', name]);
  const declared = declaredIn(this.ast.body, new Set(this.ast.params));
  walk(this.ast.body, (x) => { if (x.k === 'Assign' && x.target.k === 'Id') declared.add(x.target.name); if (x.k === 'Func') x.params.forEach((q) => declared.add(q)); });
  const callees = new Set();
  walk(this.ast.body, (x) => { if (x.k === 'Call' && x.callee.k === 'Id') callees.add(x.callee.name); });
  const missing = [name];
  for (const nm of idsOf(this.ast.body)) if (!known.has(nm) && !declared.has(nm) && !this.isStateName(nm) && nm !== this.dataName && this.freeValue(nm) === undefined) missing.push(nm);
  const fns = missing.filter((nm) => callees.has(nm)), vals = missing.filter((nm) => !callees.has(nm));
  this.fail("'" + name + "' is not defined inside log_post (free variables of the closure are invisible to the translator unless they are " +
            'globals: pass numbers/arrays in options.constants and functions in options.helpers)' +
            (missing.length > 1 || fns.length ? '; this closure needs ' + [vals.length ? 'constants: {' + vals.join(', ') + '}' : '', fns.length ? 'helpers: {' + fns.join(', ') + '}' : ''].filter(Boolean).join(', ') : ''));
};

Translator.prototype.dataValue = function (path, v) {
  if (typeof v === 'number') return cnum(v);
  if (typeof v === 'boolean') return cnum(v ? 1 : 0);
  if (typeof v === 'string') return { t: 'strlit', v };
  // categorical data: an array of strings is stored as integer codes (order of first appearance); it can only be compared
  if (Array.isArray(v) && v.length > 0 && v.every((e) => typeof e === 'string')) {
    const table = [];
    const codes = v.map((e) => { let k = table.indexOf(e); if (k < 0) { k = table.length; table.push(e); } return k; });
    const id = this.registerArray(path + '#codes', codes);
    return { t: 'strArr', id, table, n: v.length, values: v.slice() };
  }
  // an array of records ([{x: 1.2, y: 0}, ...], rows of a table): element i's field f is element i of the column f (built on demand)
  if (Array.isArray(v) && v.length > 0 && v.every((e) => e && typeof e === 'object' && !Array.isArray(e) && !ArrayBuffer.isView(e))) return { t: 'recArr', path, value: v };
  if (Array.isArray(v) || ArrayBuffer.isView(v)) {
    const id = this.registerArray(path, v);
    return { t: 'dataArr', id, off: '0', dims: this.arrays[id].dims.slice() };
  }
  if (v && typeof v === 'object') return { t: 'dataObj', path, value: v };
  this.fail('data' + path + ' is ' + (v === undefined ? 'undefined' : typeof v) + ', not a number, array or object');
};

Translator.prototype.member = function (objV, prop) {
  if (objV.t === 'stateObj') {
    if (Object.prototype.hasOwnProperty.call(this.layout, prop)) {
      const L = this.layout[prop];
      if (L.scalar) return num('S(' + L.base + ')', false);
      return { t: 'stateArr', base: String(L.base), dims: L.dim.slice() };
    }
    if (this.derived.indexOf(prop) >= 0) return num('dq_' + prop, false);
    this.fail("state." + prop + ' is read but it is neither a parameter nor a derived quantity assigned earlier');
  }
  if (objV.t === 'dataObj') {
    if (!Object.prototype.hasOwnProperty.call(objV.value, prop)) this.fail('data' + objV.path + '.' + prop + ' does not exist');
    return this.dataValue(objV.path + '.' + prop, objV.value[prop]);
  }
  if (objV.t === 'localArr') { if (prop === 'length') return cnum(objV.elems.length); this.fail("property '" + prop + "' of an array is not supported"); }
  if (objV.t === 'recArr') { if (prop === 'length') return cnum(objV.value.length); this.fail("property '" + prop + "' of an array of records is not supported"); }
  if (objV.t === 'strArr' || objV.t === 'strList') {
    if (prop === 'length') return cnum(objV.values.length);
    if (prop === 'indexOf' || prop === 'includes') return { t: 'fn', ns: 'strlist', name: prop, list: objV.values };
    this.fail("property '" + prop + "' of an array of strings is not supported");
  }
  if (objV.t === 'rec') {
    const rows = objV.arr.value, path = objV.arr.path + '[].' + prop;
    const col = rows.map((r, i) => { if (!Object.prototype.hasOwnProperty.call(r, prop)) this.fail('data' + objV.arr.path + '[' + i + '].' + prop + ' does not exist'); return r[prop]; });
    if (col.every((c) => c && typeof c === 'object' && !Array.isArray(c) && !ArrayBuffer.isView(c))) return { t: 'rec', arr: { t: 'recArr', path, value: col }, idx: objV.idx };   // nested records
    const colV = this.dataValue(path, col);
    if (colV.t !== 'dataArr' && colV.t !== 'strArr') this.fail('data' + path + ' is not a column of numbers, of strings or of equally shaped arrays');
    return this.index(colV, objV.idx);
  }
  if (objV.t === 'dataArr' || objV.t === 'stateArr') {
    if (prop === 'length') return cnum(objV.dims[0]);
    this.fail("property '" + prop + "' of an array is not supported");
  }
  if (objV.t === 'ns') {
    if (objV.name === 'Math' && Object.prototype.hasOwnProperty.call(MATH_CONST, prop)) return cnum(MATH_CONST[prop]);
    if (objV.name === 'Number' && ['EPSILON', 'MAX_VALUE', 'MIN_VALUE', 'MAX_SAFE_INTEGER', 'MIN_SAFE_INTEGER', 'POSITIVE_INFINITY', 'NEGATIVE_INFINITY', 'NaN'].indexOf(prop) >= 0) return cnum(Number[prop]);
    return { t: 'fn', ns: objV.name, name: prop };
  }
  this.fail("cannot read property '" + prop + "' of a " + this.describe(objV));
};

Translator.prototype.index = function (objV, idxV) {
  if (objV.t === 'localArr') {
    if (idxV.cst !== undefined && Number.isInteger(idxV.cst)) {
      if (idxV.cst < 0 || idxV.cst >= objV.elems.length) this.fail('constant index ' + idxV.cst + ' is outside an array of length ' + objV.elems.length);
      return objV.name ? num(objV.name + '[' + idxV.cst + ']', false) : objV.elems[idxV.cst];
    }
    // run-time index into a small local array: out of range reads give NaN, as `undefined` does in arithmetic
    const nm = this.materialize(objV), rgL = rangeOf(idxV);
    if (idxV.int && rgL && rgL[0] >= 0 && rgL[1] < objV.elems.length) return num(nm + '[' + this.asI(idxV) + ']', false);
    if (idxV.t === 'num' && !idxV.int) {      // a number that may not be an integer: undefined (NaN) unless it is one, and inside
      const dv = this.temp(this.asD(idxV));
      return num('((' + dv + ' >= 0.0 && ' + dv + ' < ' + objV.elems.length + '.0 && ' + dv + ' == __builtin_trunc(' + dv + ')) ? ' + nm + '[(int)' + dv + '] : __builtin_nan(""))', false);
    }
    const ix = this.temp_int(this.asI(idxV));
    return num('((unsigned)' + ix + ' < ' + objV.elems.length + 'u ? ' + nm + '[' + ix + '] : __builtin_nan(""))', false);
  }
  if (objV.t === 'strArr') {
    const code = this.index({ t: 'dataArr', id: objV.id, off: '0', dims: [objV.n] }, idxV);
    return code.cst !== undefined ? { t: 'strlit', v: objV.table[code.cst] } : { t: 'strv', code, table: objV.table, id: objV.id };
  }
  if (objV.t === 'recArr') {
    if (idxV.t !== 'num') this.fail('a ' + this.describe(idxV) + ' is used as an array index');
    if (idxV.cst !== undefined && (!Number.isInteger(idxV.cst) || idxV.cst < 0 || idxV.cst >= objV.value.length)) this.fail('constant index ' + idxV.cst + ' is outside an array of length ' + objV.value.length);
    if (idxV.cst !== undefined) return { t: 'rec', arr: objV, idx: idxV };
    if (idxV.int) return { t: 'rec', arr: objV, idx: withRange(num(this.temp_int(this.asI(idxV)), true), rangeOf(idxV)) };
    return { t: 'rec', arr: objV, idx: idxV };        // a non-integer-typed index: checked (range, integrality) when a field is read
  }
  if (objV.t !== 'dataArr' && objV.t !== 'stateArr') this.fail('indexing a ' + this.describe(objV));
  const dims = objV.dims, inner = dims.slice(1).reduce((a, b) => a * b, 1);
  let off;
  let guards = objV.guards || [];
  if (idxV.cst !== undefined && Number.isInteger(idxV.cst)) {
    if (idxV.cst < 0 || idxV.cst >= dims[0]) {
      if (dims.length === 1) { const u = cnum(NaN); u.undef = 'true'; return u; }      // x[n] is undefined in JavaScript: NaN in arithmetic (and === another undefined)
      this.fail('constant index ' + idxV.cst + ' is outside an array of ' + dims[0] + ' rows (JavaScript would throw on the next index)');
    }
    off = idxV.cst * inner;
  } else {
    // JavaScript reads `undefined` (NaN in arithmetic) outside the array and for a non-integer index; nothing is read from memory then.
    // Indices that provably stay inside (loop counters, elements of integer data arrays, + - * % of those) need no check.
    if (idxV.t !== 'num') this.fail('a ' + this.describe(idxV) + ' is used as an array index');
    const rg = rangeOf(idxV);
    let i;
    if (idxV.int) {
      i = this.asI(idxV);
      if (!(rg && rg[0] >= 0 && rg[1] < dims[0])) { i = this.temp_int(i); guards = guards.concat(['(unsigned)' + i + ' < ' + dims[0] + 'u']); }
    } else {
      const dv = this.temp(this.asD(idxV));
      guards = guards.concat(['(' + dv + ' >= 0.0 && ' + dv + ' < ' + dims[0] + '.0 && ' + dv + ' == __builtin_trunc(' + dv + '))']);
      i = this.temp_int('(' + guards[guards.length - 1] + ' ? (int)' + dv + ' : 0)');
    }
    off = inner === 1 ? i : '(' + i + ') * ' + inner;
  }
  const base = objV.t === 'dataArr' ? objV.off : objV.base;
  let sum;
  if (typeof off === 'number' && /^\d+$/.test(base)) sum = String(Number(base) + off);
  else if (base === '0') sum = String(off);
  else sum = base + ' + ' + off;
  const G = guards.length ? guards : undefined;
  if (dims.length > 1) return objV.t === 'dataArr' ? { t: 'dataArr', id: objV.id, off: sum, dims: dims.slice(1), guards: G } : { t: 'stateArr', base: sum, dims: dims.slice(1), guards: G };
  // (`undef`: the condition under which this read is JavaScript's `undefined` -- it matters to == / != only, see the comparison below)
  const guarded = (code) => { const v = num('((' + guards.join(' && ') + ') ? ' + code + ' : __builtin_nan(""))', false); v.undef = '!(' + guards.join(' && ') + ')'; return v; };
  if (objV.t === 'dataArr') {
    // a constant element of a data array is a constant
    if (/^\d+$/.test(sum) && !G) return cnum(this.arrays[objV.id].flat[Number(sum)]);
    const A = this.arrays[objV.id];
    if (G) return guarded((A.type === 0 ? '' : '(double)') + 'A' + objV.id + '[' + sum + ']');
    const v = A.type === 0 ? num('A' + objV.id + '[' + sum + ']', false) : num('(int)A' + objV.id + '[' + sum + ']', true, undefined, '(double)A' + objV.id + '[' + sum + ']');
    if (A.type !== 0) {        // an element of an integer array lies between the array's extremes
      if (!A.range) { let lo = Infinity, hi = -Infinity; for (let q = 0; q < A.flat.length; q++) { if (A.flat[q] < lo) lo = A.flat[q]; if (A.flat[q] > hi) hi = A.flat[q]; } A.range = A.flat.length ? [lo, hi] : null; }
      withRange(v, A.range);
    }
    v.src = { id: objV.id, off: sum };
    return v;
  }
  return G ? guarded('S(' + sum + ')') : num('S(' + sum + ')', false);
};

const ARITH = { '+': (a, b) => a + b, '-': (a, b) => a - b, '*': (a, b) => a * b, '/': (a, b) => a / b, '%': (a, b) => a % b,
  '|': (a, b) => a | b, '&': (a, b) => a & b, '^': (a, b) => a ^ b, '<<': (a, b) => a << b, '>>': (a, b) => a >> b, '>>>': (a, b) => a >>> b };
const BITOPS = { '|': 'js_bitor', '&': 'js_bitand', '^': 'js_bitxor', '<<': 'js_shl', '>>': 'js_shr', '>>>': 'js_ushr' };
const CMP = { '<': (a, b) => a < b, '<=': (a, b) => a <= b, '>': (a, b) => a > b, '>=': (a, b) => a >= b, '===': (a, b) => a === b, '==': (a, b) => a === b, '!==': (a, b) => a !== b, '!=': (a, b) => a !== b };

// Only the operands of && || ! inherit "this is a condition" from cond(); the operands of everything else are values
// (`(a || b) < c` selects a value although it stands inside a condition).
Translator.prototype.expr = function (e) {
  const saved = this.inCondition;
  if (saved && !(e.k === 'Logical' || (e.k === 'Unary' && e.op === '!'))) {
    this.inCondition = false;
    try { return this.exprInner(e, true); } finally { this.inCondition = saved; }
  }
  return this.exprInner(e, false);
};
Translator.prototype.exprInner = function (e, wasCondition) {
  switch (e.k) {
    case 'Num': return cnum(e.v);
    case 'Bool': return { t: 'bool', code: e.v ? 'true' : 'false', cst: e.v };
    case 'Id': return this.lookup(e.name);
    case 'Member': return this.member(this.expr(e.obj), e.prop);
    case 'Index': return this.index(this.expr(e.obj), this.expr(e.idx));
    case 'ArrayLit': {
      const vs = e.elems.map((x) => this.expr(x));
      if (vs.every((v) => v.t === 'num' && v.cst !== undefined)) {      // constants: a (chain-shared) data array
        const vals = vs.map((v) => v.cst);
        const id = this.registerArray('#lit' + JSON.stringify(vals), vals);
        return { t: 'dataArr', id, off: '0', dims: [vals.length] };
      }
      if (vs.length && vs.every((v) => v.t === 'strlit')) return { t: 'strList', values: vs.map((v) => v.v) };      // ['control', 'low', 'high']: only for indexOf / includes
      for (const v of vs) if (v.t !== 'num' && v.t !== 'bool') this.fail('array literals may only hold numbers (nested arrays of expressions are not supported)');
      if (vs.length > 64) this.fail('array literal with more than 64 elements');
      return { t: 'localArr', elems: vs.map((v) => num(this.asD(v), false)) };
    }
    case 'NewArray': {
      const n = this.expr(e.len);
      if (n.t !== 'num' || n.cst === undefined || !Number.isInteger(n.cst) || n.cst < 1 || n.cst > 2048)
        this.fail('the length of a local array (Array(n), map()) must be a constant between 1 and 2048 known when the sampler is built' + (n.cst > 2048 ? ' (got ' + n.cst + ': every lane would carry its own copy; write the loop with a scalar temporary instead)' : ''));
      const fill = e.fill ? this.expr(e.fill) : cnum(NaN);          // holes read as undefined -> NaN in arithmetic
      if (fill.t !== 'num' && fill.t !== 'bool') this.fail('Array(n).fill(v) needs a number');
      const one = num(this.asD(fill), false);
      return { t: 'localArr', elems: Array.from({ length: n.cst }, () => one), fillOnly: true };
    }
    case 'Unary': {
      const a = this.expr(e.arg);
      if (e.op === '!') { if (a.cst !== undefined) return { t: 'bool', code: a.cst ? 'false' : 'true', cst: !a.cst }; return { t: 'bool', code: '!(' + this.asB(a) + ')' }; }
      if (a.t !== 'num') this.fail("unary '" + e.op + "' on a " + this.describe(a));
      if (e.op === '+') return a;
      if (e.op === '~') return a.cst !== undefined ? cnum(~a.cst) : num('js_bitnot(' + this.asD(a) + ')', false);
      if (a.cst !== undefined) return cnum(-a.cst);
      return a.int ? withRange(num('(-(' + a.code + '))', true, undefined, '(-(' + a.dcode + '))'), rangeOf(a) && [-rangeOf(a)[1], -rangeOf(a)[0]]) : num('(-(' + a.code + '))', false);
    }
    case 'Binary': {
      const l = this.expr(e.l), r = this.expr(e.r);
      if (CMP[e.op] && (l.t === 'strlit' || l.t === 'strv' || r.t === 'strlit' || r.t === 'strv')) {
        // categorical values: equality only, as integer codes
        const eq = e.op === '===' || e.op === '==', ne = e.op === '!==' || e.op === '!=';
        if (!eq && !ne) this.fail("strings can only be compared with === / !==, not '" + e.op + "'");
        const isStr = (v) => v.t === 'strlit' || v.t === 'strv';
        if (!isStr(l) || !isStr(r)) {       // a string against a number / boolean: never equal under === (== would coerce: refused)
          if (e.op === '==' || e.op === '!=') this.fail("'" + e.op + "' between a string and a " + this.describe(isStr(l) ? r : l) + ' (use === )');
          return { t: 'bool', code: ne ? 'true' : 'false', cst: ne };
        }
        if (l.t === 'strlit' && r.t === 'strlit') { const v = (l.v === r.v) === eq; return { t: 'bool', code: v ? 'true' : 'false', cst: v }; }
        if (l.t === 'strv' && r.t === 'strv') {
          if (l.id !== r.id) this.fail('comparing elements of two different arrays of strings is not supported');
          return { t: 'bool', code: '(' + l.code.code + (eq ? ' == ' : ' != ') + r.code.code + ')' };
        }
        const sv = l.t === 'strv' ? l : r, lit = l.t === 'strv' ? r : l, k = sv.table.indexOf(lit.v);
        if (k < 0) return { t: 'bool', code: ne ? 'true' : 'false', cst: ne };      // a label that never occurs
        return { t: 'bool', code: '(' + sv.code.code + (eq ? ' == ' : ' != ') + k + ')' };
      }
      if (CMP[e.op]) {
        if (l.t === 'bool' && r.t === 'bool') return { t: 'bool', code: '((' + l.code + ') ' + (e.op[0] === '!' ? '!=' : '==') + ' (' + r.code + '))' };
        if (l.t !== 'num' || r.t !== 'num') this.fail("comparison '" + e.op + "' between a " + this.describe(l) + ' and a ' + this.describe(r));
        // two reads outside their arrays are both `undefined`, and undefined == undefined (=== too) although each is NaN in arithmetic
        if (l.undef && r.undef && (e.op === '==' || e.op === '===' || e.op === '!=' || e.op === '!==')) {
          const both = (l.undef === 'true' && r.undef === 'true') ? 'true' : '((' + l.undef + ') && (' + r.undef + '))';
          const eqc = '((' + this.asD(l) + ' == ' + this.asD(r) + ') || ' + both + ')';
          return { t: 'bool', code: e.op[0] === '!' ? '(!' + eqc + ')' : eqc };
        }
        if (l.cst !== undefined && r.cst !== undefined) { const v = CMP[e.op](l.cst, r.cst); return { t: 'bool', code: v ? 'true' : 'false', cst: v }; }
        const op = e.op === '===' ? '==' : (e.op === '!==' ? '!=' : e.op);
        // loop counters against integer bounds compare as ints; everything else as JavaScript numbers
        if (l.int && r.int && (l.cst !== undefined || r.cst !== undefined || (/^v_\w+$/.test(l.code) && /^v_\w+$/.test(r.code)))) return { t: 'bool', code: '(' + l.code + ' ' + op + ' ' + r.code + ')' };
        return { t: 'bool', code: '(' + this.asD(l) + ' ' + op + ' ' + this.asD(r) + ')' };
      }
      if (l.t !== 'num' || r.t !== 'num') {
        if (e.op === '+' ) this.fail("'+' between a " + this.describe(l) + ' and a ' + this.describe(r) + ' (string concatenation is not supported)');
        this.fail("'" + e.op + "' between a " + this.describe(l) + ' and a ' + this.describe(r));
      }
      if (l.cst !== undefined && r.cst !== undefined) return cnum(ARITH[e.op](l.cst, r.cst));   // folded by V8 itself
      if (e.op === '/') return num('(' + this.asD(l) + ' / ' + this.asD(r) + ')', false);
      if (BITOPS[e.op]) return num(BITOPS[e.op] + '(' + this.asD(l) + ', ' + this.asD(r) + ')', false);     // ToInt32 of both operands, as in JS
      if (e.op === '%') {
        if (l.int && r.int && r.cst !== undefined && r.cst !== 0) {
          const a = rangeOf(l), c = Math.abs(r.cst);      // the sign of % follows the dividend
          const rg = a && a[0] >= 0 ? [0, Math.min(a[1], c - 1)] : [-(c - 1), c - 1];
          return withRange(num('(' + l.code + ' % ' + r.code + ')', true, undefined, 'js_mod(' + this.asD(l) + ', ' + this.asD(r) + ')'), rg);
        }
        return num('js_mod(' + this.asD(l) + ', ' + this.asD(r) + ')', false);
      }
      if (l.int && r.int) {
        const span = (a, b) => {
          if (e.op === '+') return [a[0] + b[0], a[1] + b[1]];
          if (e.op === '-') return [a[0] - b[1], a[1] - b[0]];
          const c = [a[0] * b[0], a[0] * b[1], a[1] * b[0], a[1] * b[1]];
          return [Math.min.apply(null, c), Math.max.apply(null, c)];
        };
        const a = rangeOf(l), b = rangeOf(r);
        // 32-bit int arithmetic is only used while it cannot overflow (values of unknown range -- counters of loops whose bounds are not
        // constants -- are taken to stay below 2^30); otherwise the operation is done on doubles, exactly as JavaScript does it
        const W = [-1073741824, 1073741824], wide = span(a || W, b || W);
        if (!(wide[0] > -2147483648 && wide[1] < 2147483648)) return num('(' + this.asD(l) + ' ' + e.op + ' ' + this.asD(r) + ')', false);
        return withRange(num('(' + l.code + ' ' + e.op + ' ' + r.code + ')', true, undefined, '(' + this.asD(l) + ' ' + e.op + ' ' + this.asD(r) + ')'), a && b ? span(a, b) : null);
      }
      return num('(' + this.asD(l) + ' ' + e.op + ' ' + this.asD(r) + ')', false);
    }
    case 'Logical': {
      const l = this.expr(e.l), r = this.expr(e.r);
      if ((l.t !== 'bool' && l.t !== 'num') || (r.t !== 'bool' && r.t !== 'num')) this.fail("'" + e.op + "' on a " + this.describe(l.t !== 'bool' ? l : r));
      if ((l.t === 'num' || r.t === 'num') && !this.inCondition) {
        // value-selecting `a || b` / `a && b` on numbers (var n = opts.n || 10): a number is falsy when it is 0, -0 or NaN
        if (l.t !== 'num' || r.t !== 'num') this.fail("'" + e.op + "' between a number and a condition is only supported inside conditions");
        if (l.cst !== undefined) { const truthy = l.cst !== 0 && l.cst === l.cst; return (e.op === '||') === truthy ? l : r; }
        const a = this.temp(this.asD(l)), tr = '(' + a + ' != 0.0 && ' + a + ' == ' + a + ')';
        return num('(' + tr + ' ? ' + (e.op === '||' ? a + ' : ' + this.asD(r) : this.asD(r) + ' : ' + a) + ')', false);
      }
      return { t: 'bool', code: '(' + this.asB(l) + ' ' + e.op + ' ' + this.asB(r) + ')' };
    }
    case 'Cond': {
      const t = this.cond(e.test), a = this.expr(e.a), b = this.expr(e.b);
      if (t.cst !== undefined) return t.cst ? a : b;
      if (a.t === 'bool' && b.t === 'bool') return { t: 'bool', code: '(' + t.code + ' ? ' + a.code + ' : ' + b.code + ')' };
      if (a.t === 'strlit' && b.t === 'strlit') {       // c ? 'x' : 'z': a categorical value over the two labels
        if (a.v === b.v) return a;
        return { t: 'strv', code: withRange(num('(' + t.code + ' ? 0 : 1)', true), [0, 1]), table: [a.v, b.v], id: 'cond:' + (this.tmp++) };
      }
      if (a.t !== 'num' || b.t !== 'num') this.fail('both branches of ?: must be numbers (or both string literals)');
      if (a.int && b.int) return withRange(num('(' + t.code + ' ? ' + a.code + ' : ' + b.code + ')', true, undefined, '(' + t.code + ' ? ' + this.asD(a) + ' : ' + this.asD(b) + ')'),
        rangeOf(a) && rangeOf(b) && [Math.min(rangeOf(a)[0], rangeOf(b)[0]), Math.max(rangeOf(a)[1], rangeOf(b)[1])]);
      return num('(' + t.code + ' ? ' + this.asD(a) + ' : ' + this.asD(b) + ')', false);
    }
    case 'Call': return this.call(e);
    case 'Func': this.fail('a function expression can only be assigned to a variable (var f = function (x) {...})');   // eslint-disable-line no-fallthrough
    case 'Assign': case 'Update': this.fail('assignments are only supported as statements');  // eslint-disable-line no-fallthrough
    case 'Str': return { t: 'strlit', v: e.v };      // only ever compared (=== !== == !=) with categorical data
    case 'Seq': this.fail("the ',' operator is not supported");   // eslint-disable-line no-fallthrough
  }
  this.fail('unsupported expression (' + e.k + ')');
};

Translator.prototype.cond = function (e) {
  const saved = this.inCondition;
  this.inCondition = true;
  const v = this.expr(e);
  this.inCondition = saved;
  if (v.t === 'bool') return v;
  return { t: 'bool', code: this.asB(v), cst: v.cst !== undefined ? !!v.cst : undefined };
};

// Pure calls (Math.exp/log/pow/sqrt, ld.*, helpers) whose arguments cannot change inside the enclosing loop(s) are
// evaluated once, just before the outermost such loop -- the same value, computed by the same code.
Translator.prototype.call = function (e) {
  const v = this.callInner(e);
  if (v.t !== 'num' || v.cst !== undefined || v.int || !this.loops.length || this.noHoist || this.pending.length) return v;
  if (!/^(exp_v8|log_v8|pow_v8|log1p_v8|log1p_exp_v8|expm1_v8|tanh_v8|atan_v8|log10_v8|__builtin_sqrt|ld_\w+|lgamma_js|lfactorial_js|lchoose_js|lbeta_js|h_\w+)\(/.test(v.code)) return v;
  if (/NORMCALL|_inv\(|_inv01\(|_pre\(/.test(v.code)) return v;            // already specialised for the loop
  const k = this.hoist(e, 'double', '', v.code, '');
  return k ? num(k, false) : v;
};

Translator.prototype.callInner = function (e) {
  const f = this.expr(e.callee);
  if (f.t !== 'fn') this.fail('calling a ' + this.describe(f));
  const args = e.args.map((a) => this.expr(a));
  const nums = () => args.map((a) => { if (a.t !== 'num' && a.t !== 'bool') this.fail(f.ns + '.' + f.name + ' got a ' + this.describe(a) + ' argument (only scalar arguments are supported)'); return a; });
  const allConst = () => args.every((a) => a.cst !== undefined && a.t === 'num');
  if (f.ns === 'strlist') {
    // LEVELS.indexOf(row.group) / LEVELS.includes(row.group): the position of a categorical value in a constant list of labels, through a
    // table from the column's integer codes to positions (built here, once)
    if (args.length !== 1) this.fail(f.name + ' takes one argument here');
    const a = args[0];
    if (a.t === 'strlit') { const k = f.list.indexOf(a.v); return f.name === 'indexOf' ? cnum(k) : { t: 'bool', code: k >= 0 ? 'true' : 'false', cst: k >= 0 }; }
    if (a.t !== 'strv') { if (f.name === 'includes') return { t: 'bool', code: 'false', cst: false }; return cnum(-1); }     // a number is never === a string
    const map = a.table.map((label) => f.list.indexOf(label));
    const tab = this.dataValue('#indexOf:' + JSON.stringify([a.id, f.list]), map);
    const pos = this.index(tab, a.code);
    if (f.name === 'indexOf') return pos;
    return { t: 'bool', code: '(' + this.asD(pos) + ' >= 0.0)' };
  }
  if (f.ns === 'Number' && (f.name === 'isInteger' || f.name === 'isSafeInteger')) {
    if (args.length !== 1 || (args[0].t !== 'num' && args[0].t !== 'bool')) this.fail('Number.' + f.name + ' takes one number');
    if (args[0].t === 'bool') return { t: 'bool', code: 'false', cst: false };
    if (args[0].cst !== undefined) { const r = Number[f.name](args[0].cst); return { t: 'bool', code: r ? 'true' : 'false', cst: r }; }
    const x = this.temp(this.asD(args[0]));
    return { t: 'bool', code: '(__builtin_fabs(' + x + ') < ' + (f.name === 'isInteger' ? 'kInf' : '9007199254740992.0') + ' && __builtin_trunc(' + x + ') == ' + x + ')' };
  }
  if (f.ns === 'global' || (f.ns === 'Number' && (f.name === 'isNaN' || f.name === 'isFinite'))) {
    if (args.length !== 1 || args[0].t !== 'num') this.fail(f.name + ' takes one number');
    if (args[0].cst !== undefined) { const r = f.name === 'isNaN' ? Number.isNaN(args[0].cst) : Number.isFinite(args[0].cst); return { t: 'bool', code: r ? 'true' : 'false', cst: r }; }
    const x = this.temp(this.asD(args[0]));
    return { t: 'bool', code: f.name === 'isNaN' ? '(' + x + ' != ' + x + ')' : '(__builtin_fabs(' + x + ') < kInf)' };
  }
  if (f.ns === 'Math') {
    nums();
    if (f.name === 'pow') {
      if (args.length !== 2) this.fail('Math.pow takes two arguments');
      if (allConst()) return cnum(Math.pow(args[0].cst, args[1].cst));
      const x = this.asD(args[0]);
      // V8's pow returns exactly x*x for an exponent of 2 (fdlibm e_pow.c special case), sqrt(x) for 0.5 and x >= 0
      if (args[1].cst === 2) { const t = this.temp(x); return num('(' + t + ' * ' + t + ')', false, undefined); }
      if (args[1].cst === 1) return num(x, false);
      if (this.loops.length) this.heavyLoop = true;
      return num('pow_v8(' + x + ', ' + this.asD(args[1]) + ')', false);
    }
    if (f.name === 'min' || f.name === 'max') {
      if (args.length < 1) this.fail('Math.' + f.name + ' needs arguments');
      if (allConst()) return cnum(Math[f.name].apply(null, args.map((a) => a.cst)));
      let code = this.asD(args[0]);
      for (let k = 1; k < args.length; k++) code = 'js_' + f.name + '(' + code + ', ' + this.asD(args[k]) + ')';
      return num(code, false);
    }
    if (f.name === 'imul') {
      if (args.length !== 2) this.fail('Math.imul takes two arguments');
      if (allConst()) return cnum(Math.imul(args[0].cst, args[1].cst));
      return num('js_imul(' + this.asD(args[0]) + ', ' + this.asD(args[1]) + ')', false);
    }
    if (f.name === 'atan2') {
      if (args.length !== 2) this.fail('Math.atan2 takes two arguments');
      if (allConst()) return cnum(Math.atan2(args[0].cst, args[1].cst));
      if (this.loops.length) this.heavyLoop = true;
      return num('atan2_v8(' + this.asD(args[0]) + ', ' + this.asD(args[1]) + ')', false);
    }
    if (f.name === 'hypot') {
      if (args.length < 1 || args.length > 4) this.fail('Math.hypot takes 1 to 4 arguments here');
      if (allConst()) return cnum(Math.hypot.apply(null, args.map((a) => a.cst)));
      if (this.loops.length) this.heavyLoop = true;
      if (args.length === 1) return num('hypot2_v8(' + this.asD(args[0]) + ', 0.0)', false);     // sqrt(1) * |x|
      return num('hypot' + args.length + '_v8(' + args.map((a) => this.asD(a)).join(', ') + ')', false);
    }
    const M = MATH_FUNS[f.name];
    if (!M) this.fail('Math.' + f.name + ' is not supported');
    if (args.length !== M[1]) this.fail('Math.' + f.name + ' takes ' + M[1] + ' argument(s)');
    if (allConst()) return cnum(M[2](args[0].cst));
    if (HEAVY.has(M[0]) && this.loops.length) this.heavyLoop = true;
    if ((f.name === 'floor' || f.name === 'ceil' || f.name === 'round' || f.name === 'trunc' || f.name === 'abs') && args[0].int)
      return f.name === 'abs' ? num('(' + args[0].code + ' < 0 ? -(' + args[0].code + ') : ' + args[0].code + ')', true, undefined, '__builtin_fabs(' + args[0].dcode + ')') : args[0];
    // Math.log1p(Math.exp(eta)) -- softplus, the logistic log-likelihood -- is ONE straight-line function on the device (csrc/amwg_math.h
    // log1p_exp_v8: the branches of fdlibm's log1p as selects; same bits as log1p_v8(exp_v8(eta)))
    if (f.name === 'log1p' && args[0].expOf !== undefined) return num('log1p_exp_v8(' + args[0].expOf + ')', false);
    const r = num(M[0] + '(' + this.asD(args[0]) + ')', false);
    if (f.name === 'exp') r.expOf = this.asD(args[0]);      // (ld.pois(x, Math.exp(eta)) fuses the two, see below)
    return r;
  }
  if (f.ns === 'ld' && (f.name === 'dirichlet' || f.name === 'cat' || f.name === 'bivarnorm')) return this.arrayDensity(f.name, args);
  if (f.ns === 'ld') {
    const L = LD_FUNS[f.name];
    if (!L) this.fail('ld.' + f.name + ' does not exist in distributions.js');
    if (args.length !== L[1]) this.fail('ld.' + f.name + ' takes ' + L[1] + ' arguments, got ' + args.length);
    nums();
    const a = args.map((x) => this.asD(x));
    if (f.name === 'norm') {
      // NORMCALL / RANGEARGS / KFASTCHECK are resolved when the enclosing loop is finished (renderNorm)
      const k = this.hoist(e.args[2], 'NormInv', 'norm_inv', a[2], ' KFASTCHECK(@)');
      if (k) return num('NORMCALL(' + a[0] + ', ' + a[1] + ', ' + k + ' RANGEARGS)', false);
    }
    if (f.name === 'bern') {
      const k = this.hoist(e.args[1], 'BernInv', 'bern_inv', a[1], '');
      if (k) {
        if (args[0].src && this.arrays[args[0].src.id].is01) return num('ld_bern_inv01(A' + args[0].src.id + '[' + args[0].src.off + '], ' + k + ')', false);
        return num('ld_bern_inv(' + a[0] + ', ' + k + ')', false);
      }
    }
    if (f.name === 'pois' && args[0].src && this.loops.length) {      // lfactorial(x_i) depends on the data only: once, on the host
      const aux = this.auxArray('lfactorial', [args[0].src.id], (x) => ld_host.lfactorial(x));
      this.heavyLoop = true;
      if (args[1].expOf) return num('ld_pois_pre_exp(' + a[0] + ', ' + args[1].expOf + ', A' + aux + '[' + args[0].src.off + '])', false);
      return num('ld_pois_pre(' + a[0] + ', ' + a[1] + ', A' + aux + '[' + args[0].src.off + '])', false);
    }
    if (f.name === 'pois' && args[1].expOf) { if (this.loops.length) this.heavyLoop = true; return num('ld_pois_exp(' + a[0] + ', ' + args[1].expOf + ')', false); }
    if (f.name === 'binom' && args[0].src && args[1].src && args[0].src.off === args[1].src.off && this.loops.length &&
        this.arrays[args[0].src.id].flat.length === this.arrays[args[1].src.id].flat.length) {
      const aux = this.auxArray('lchoose', [args[1].src.id, args[0].src.id], (size, x) => ld_host.lchoose(size, x));
      this.heavyLoop = true;
      return num('ld_binom_pre(' + a[0] + ', ' + a[1] + ', ' + a[2] + ', A' + aux + '[' + args[0].src.off + '])', false);
    }
    if (this.loops.length) this.heavyLoop = true;
    return num(L[0] + '(' + a.join(', ') + ')', false);
  }
  if (f.ns === 'helper') {
    const h = this.helper(f.name);
    if (args.length !== h.nargs) this.fail('helper ' + f.name + ' takes ' + h.nargs + ' argument(s)');
    nums();
    if (this.loops.length) this.heavyLoop = true;
    return num(h.cname + '(' + args.map((x) => this.asD(x)).join(', ') + ')', false);
  }
  this.fail('unsupported call');
};

// data-only part of a density, evaluated per observation on the host with the reference's own formula (ld.js)
Translator.prototype.auxArray = function (fname, ids, f) {
  const key = '#aux:' + fname + ':' + ids.join(',');
  if (this.arrayIds.has(key)) return this.arrayIds.get(key);
  const n = this.arrays[ids[0]].flat.length, vals = new Array(n);
  for (let i = 0; i < n; i++) vals[i] = f.apply(null, ids.map((id) => this.arrays[id].flat[i]));
  return this.registerArray(key, vals);
};

Translator.prototype.temp_int = function (code) {
  if (/^[\w]+$/.test(code)) return code;
  const name = 'ti' + (this.tmp++);
  this.pending.push('const int ' + name + ' = ' + code + ';');
  return name;
};

// an array of expressions as a C array (needed when it is indexed with a run-time value)
Translator.prototype.materialize = function (arr) {
  if (arr.name) return arr.name;
  const name = 'la' + (this.tmp++);
  this.pending.push('const double ' + name + '[' + arr.elems.length + '] = {' + arr.elems.map((v) => v.code).join(', ') + '};');
  return name;
};

// the elements of a one-dimensional array value (any kind) as numbers
Translator.prototype.elementsOf = function (v, what) {
  if (v.t === 'localArr') return v.name ? v.elems.map((_, i) => num(v.name + '[' + i + ']', false)) : v.elems;
  if ((v.t === 'dataArr' || v.t === 'stateArr') && v.dims.length === 1) {
    if (v.dims[0] > 64) this.fail(what + ': arrays longer than 64 are not supported here');
    const out = [];
    for (let i = 0; i < v.dims[0]; i++) out.push(this.index(v, cnum(i)));
    return out;
  }
  this.fail(what + ' needs a one-dimensional array, got a ' + this.describe(v));
};

// The six data-only tables of csrc/amwg_twoval.h for a 0/1 array, back to back, as signed 32-bit words (i32 storage).
Translator.prototype.twoValuedTables = function (id) {
  const key = '#aux:twoval:' + id;
  if (this.arrayIds.has(key)) return this.arrayIds.get(key);
  const x = this.arrays[id].flat, N = x.length, W = Math.floor(N / 32) + 2;
  const tab = new Uint32Array(6 * W);
  for (let i = 0; i < N; i++) if (x[i] === 1) tab[i >> 5] |= (1 << (i & 31)) >>> 0;
  const popc = (v) => { v = v - ((v >>> 1) & 0x55555555); v = (v & 0x33333333) + ((v >>> 2) & 0x33333333); return (((v + (v >>> 4)) & 0x0f0f0f0f) * 0x01010101) >>> 24; };
  for (let k = 1; k < W; k++) tab[W + k] = tab[W + k - 1] + popc(tab[k - 1]);
  for (const sym of [1, 0]) {
    const om = (sym ? 2 : 4) * W, po = om + W;
    let run = 0;
    for (let i = 0; i < N; i++) {
      if (x[i] === sym) { if (run & 1) tab[om + (i >> 5)] |= (1 << (i & 31)) >>> 0; run = 0; } else run++;
    }
    for (let k = 1; k < W; k++) tab[po + k] = tab[po + k - 1] + popc(tab[om + k - 1]);
  }
  const signed = new Array(6 * W);
  for (let k = 0; k < 6 * W; k++) signed[k] = tab[k] | 0;
  const out = this.registerArray(key, signed);
  if (this.arrays[out].type === 1) { this.arrays[out].type = 2; this.arrays[out].ctype = 'int32_t'; this.arrays[out].esize = 4; this.arrays[out].is01 = false; }   // always i32
  return out;
};

// The data-only tables of csrc/amwg_kval.h for an array with at most 16 distinct values (a count column, ...): per distinct value its
// occurrence mask and prefix counts, back to back as 32-bit words, plus the value INDEX of every observation as its own (byte) array.
// -> {tab, idx, first: [index of the first occurrence of every distinct value], K} or null when the array has too many distinct values
Translator.prototype.kValuedTables = function (id) {
  const key = '#aux:kval:' + id;
  if (this.kvalCache && this.kvalCache[key]) return this.kvalCache[key];
  const x = this.arrays[id].flat, N = x.length, W = Math.floor(N / 32) + 2;
  const slot = new Map(), first = [];
  for (let i = 0; i < N; i++) {
    if (!slot.has(x[i])) { if (first.length === 16) return null; slot.set(x[i], first.length); first.push(i); }
  }
  const K = first.length;
  if (K < 1 || N < 64) return null;
  const tab = new Uint32Array(2 * K * W), idx = new Array(N);
  // per block of 32 observations: pre[K] then mask[K], side by side (csrc/amwg_kval.h)
  for (let i = 0; i < N; i++) { const k = slot.get(x[i]); idx[i] = k; tab[(i >> 5) * 2 * K + K + k] |= (1 << (i & 31)) >>> 0; }
  const popc = (v) => { v = v - ((v >>> 1) & 0x55555555); v = (v & 0x33333333) + ((v >>> 2) & 0x33333333); return (((v + (v >>> 4)) & 0x0f0f0f0f) * 0x01010101) >>> 24; };
  for (let w = 1; w < W; w++) for (let k = 0; k < K; k++) tab[w * 2 * K + k] = tab[(w - 1) * 2 * K + k] + popc(tab[(w - 1) * 2 * K + K + k]);
  const signed = new Array(2 * K * W);
  for (let k = 0; k < 2 * K * W; k++) signed[k] = tab[k] | 0;
  const t = this.registerArray(key + ':tab', signed);
  if (this.arrays[t].type === 1) { this.arrays[t].type = 2; this.arrays[t].ctype = 'int32_t'; this.arrays[t].esize = 4; this.arrays[t].is01 = false; }   // always i32
  const j = this.registerArray(key + ':idx', idx);
  if (this.arrays[j].type !== 1) return null;      // (value indices 0 .. 15 are stored as bytes)
  this.kvalCache = this.kvalCache || {};
  return (this.kvalCache[key] = { tab: t, idx: j, first, K });
};

// a named temporary for a value that is used twice (keeps the evaluation single, as in JS)
Translator.prototype.temp = function (code) {
  if (/^[\w.]+$/.test(code) || /^S\(\d+\)$/.test(code)) return code;
  const name = 't' + (this.tmp++);
  this.pending.push('const double ' + name + ' = ' + code + ';');
  return name;
};

// Hoists `ctor(code)` to just before the outermost enclosing loop in which argAst cannot change.
Translator.prototype.hoist = function (argAst, type, ctor, code, suffix) {
  if (this.noHoist || this.loops.length === 0) return null;
  const free = idsOf(argAst);
  let j = this.loops.length;
  while (j > 0) {
    const L = this.loops[j - 1];
    let variant = false;
    for (const nm of free) if (L.assigned.has(nm)) { variant = true; break; }
    if (variant) break;
    j--;
  }
  if (j === this.loops.length) return null;   // changes inside the innermost loop
  if (this.pending.length) return null;       // the argument needs temporaries computed inside the loop
  const name = 'k' + (this.tmp++);
  this.loops[j].preamble.push('const ' + type + ' ' + name + ' = ' + ctor + '(' + code + ');' + (suffix || '').replace('@', name));
  return name;
};

// ld.dirichlet / ld.cat / ld.bivarnorm (distributions.js:125-134, 203-214, 232-238): the loops of the reference are
// unrolled over the (translation-time) lengths, every operation in the reference's order.
Translator.prototype.arrayDensity = function (name, args) {
  if (this.loops.length) this.heavyLoop = true;
  const T = (code) => this.temp(code);
  if (name === 'dirichlet') {
    if (args.length !== 2) this.fail('ld.dirichlet takes (x, alpha)');
    const x = this.elementsOf(args[0], 'ld.dirichlet'), al = this.elementsOf(args[1], 'ld.dirichlet');
    if (x.length < al.length) this.fail('ld.dirichlet: x is shorter than alpha');
    let sa = '0.0', sl = '0.0', sx = '0.0';
    for (let i = 0; i < al.length; i++) {
      const a = T(this.asD(al[i]));
      sa = T('(' + sa + ' + ' + a + ')');
      sl = T('(' + sl + ' + lgamma_js(' + a + '))');
      sx = T('(' + sx + ' + ((' + a + ' - 1.0) * log_v8(' + this.asD(x[i]) + ')))');
    }
    return num('((lgamma_js(' + sa + ') - ' + sl + ') + ' + sx + ')', false);
  }
  if (name === 'cat') {
    if (args.length !== 2) this.fail('ld.cat takes (x, probs)');
    if (args[0].t !== 'num') this.fail('ld.cat: x must be a number');
    const n = args[1].t === 'localArr' ? args[1].elems.length : ((args[1].dims || [])[0]);
    if (n === undefined || (args[1].dims && args[1].dims.length !== 1)) this.fail('ld.cat: probs must be a one-dimensional array');
    const x = T(this.asD(args[0]));
    const pr = this.index(args[1], num('(int)(' + x + ') - 1', true));   // probs[x - 1]; only evaluated inside the range below
    return num('((' + x + ' < 1.0 || ' + x + ' > ' + n + '.0) ? -kInf : log_v8(' + this.asD(pr) + '))', false);
  }
  // bivarnorm(x, mean, sd, corr)
  if (args.length !== 4) this.fail('ld.bivarnorm takes (x, mean, sd, corr)');
  const x = this.elementsOf(args[0], 'ld.bivarnorm'), m = this.elementsOf(args[1], 'ld.bivarnorm'), sd = this.elementsOf(args[2], 'ld.bivarnorm');
  if (x.length < 2 || m.length < 2 || sd.length < 2 || args[3].t !== 'num') this.fail('ld.bivarnorm takes two-element x, mean, sd and a number corr');
  const c = T(this.asD(args[3])), s0 = T(this.asD(sd[0])), s1 = T(this.asD(sd[1]));
  const d0 = T('(' + this.asD(x[0]) + ' - ' + this.asD(m[0]) + ')'), d1 = T('(' + this.asD(x[1]) + ' - ' + this.asD(m[1]) + ')');
  const z = T('((((' + d0 + ' * ' + d0 + ') / (' + s0 + ' * ' + s0 + ')) + ((' + d1 + ' * ' + d1 + ') / (' + s1 + ' * ' + s1 + '))) - ((((2.0 * ' + c + ') * ' + d0 + ') * ' + d1 + ') / (' + s0 + ' * ' + s1 + ')))');
  const nf = T('(-((((log_v8(2.0) + log_v8(kPi)) + log_v8(' + s0 + ')) + log_v8(' + s1 + ')) + (0.5 * log_v8(1.0 - (' + c + ' * ' + c + ')))))');
  return num('(' + nf + ' - (' + z + ' / (2.0 * (1.0 - (' + c + ' * ' + c + ')))))', false);
};

// ---- helper functions (options.helpers) --------------------------------------------------------
Translator.prototype.helper = function (name) {
  if (this.helpers[name]) return this.helpers[name];
  const fn = (this.localFuncs && this.localFuncs[name]) || (this.opts.helpers || {})[name];
  if (!(typeof fn === 'function' || (fn && fn.k === 'Func'))) this.fail('options.helpers.' + name + ' is not a function');
  const sub = new Translator(fn, {}, null, { constants: this.opts.constants, helpers: this.opts.helpers }, true);
  sub.helpers = this.helpers;
  sub.localFuncs = this.localFuncs;
  sub.arrays = this.arrays; sub.arrayIds = this.arrayIds;
  sub.helperSources = this.helperSources;
  const h = { cname: 'h_' + name, nargs: sub.ast.params.length };
  this.helpers[name] = h;
  const body = sub.functionBody(sub.ast.params, false);
  this.helperSources.push('AMWG_HD double ' + h.cname + '(' + sub.ast.params.map((p) => 'double v_' + p).join(', ') + ') {\n' + body.join('\n') + '\n}');
  return h;
};

// ---- statements ----------------------------------------------------------------------------------
Translator.prototype.setLocal = function (name, v) {     // records the inferred C++ type of a local
  if (v.t === 'bool') v = num(this.asD(v), false);
  if (v.t !== 'num') return false;
  // int only for the counters of canonical for loops (array indices); every other local is a JavaScript number
  if (!Object.prototype.hasOwnProperty.call(this.localTypes, name)) this.localTypes[name] = (v.int && this.loopCounters.has(name)) ? 'int' : 'double';
  else if (this.localTypes[name] === 'int' && !v.int) { this.localTypes[name] = 'double'; this.retype = true; }
  if (this.forcedDouble.has(name) && this.localTypes[name] === 'int') this.localTypes[name] = 'double';
  return true;
};

Translator.prototype.flush = function (out, indent) { for (const p of this.pending) out.push(indent + p); this.pending = []; };

Translator.prototype.assign = function (target, op, valueAst, out, indent, ctx) {
  if (target.k === 'Id') {
    const name = target.name;
    if (this.isStateName(name) || (this.dataName && name === this.dataName)) this.fail('assigning to ' + name);
    if (valueAst.k === 'Func') {
      // `var f = function (a, b) {...}` / an arrow: a helper that may use its arguments, constants and other helpers (a closure over
      // the surrounding locals is not supported: the translator would have to capture their values)
      if (op !== '=' || this.loops.length || this.condDepth) this.fail('function ' + name + ' must be defined by a plain assignment at the top level of log_post');
      this.localFuncs = this.localFuncs || {};
      this.localFuncs[name] = valueAst;
      return;
    }
    if (this.linear && this.accSet.has(name)) {
      if (Object.prototype.hasOwnProperty.call(this.aliases, name)) this.fail(name + ' holds an array/object elsewhere and a number here');
      this.setLocal(name, num('', false));
      const v_ = 'v_' + name;
      if (this.loops.length) {          // inside a loop: += / -= of an accumulator-free term (linearAccumulators)
        const t = this.expr(valueAst);
        this.flush(out, indent);
        out.push(indent + (ctx.split ? '' : 'if (sub == 0) ') + v_ + ' ' + op + ' ' + this.asD(t) + ';');
        return;
      }
      const lc = this.linearCode(valueAst);
      this.flush(out, indent);
      if (op === '=') out.push(indent + v_ + ' = ' + (lc.hasA ? lc.code : '(sub == 0) ? ' + lc.code + ' : 0.0') + ';');
      else out.push(indent + (lc.hasA ? '' : 'if (sub == 0) ') + v_ + ' ' + op + ' ' + lc.code + ';');
      return;
    }
    let v = this.expr(valueAst);
    if (v.t === 'localArr' && op === '=' && v.name && !this.loops.length && !this.condDepth && !Object.prototype.hasOwnProperty.call(this.localTypes, name)) {
      this.aliases[name] = v;        // `var z = otherArray`: arrays are references in JavaScript, both names mean the same storage
      return;
    }
    if (v.t === 'localArr' && op === '=') {
      // a variable holding an array of numbers: C array declared at the top, filled here (elements may be reassigned later)
      if (Object.prototype.hasOwnProperty.call(this.localTypes, name)) this.fail(name + ' holds a number elsewhere and an array here');
      const prev = this.aliases[name];
      if (prev && !(prev.t === 'localArr' && prev.elems.length === v.elems.length)) this.fail(name + ' is re-assigned an array of a different length');
      const cname = 'va_' + name;
      this.localArrays[cname] = v.elems.length;
      this.flush(out, indent);
      if (v.fillOnly && v.elems.length > 8) { const f = this.temp(v.elems[0].code); out.push(indent + 'for (int fi_ = 0; fi_ < ' + v.elems.length + '; ++fi_) ' + cname + '[fi_] = ' + f + ';'); }
      else v.elems.forEach((el, i) => out.push(indent + cname + '[' + i + '] = ' + el.code + ';'));
      this.aliases[name] = { t: 'localArr', elems: v.elems, name: cname };
      return;
    }
    if (v.t !== 'num' && v.t !== 'bool') {
      if (op !== '=') this.fail("'" + op + "' with a " + this.describe(v));
      if (this.loops.length || this.condDepth) {
        // `var row = data[i]` / `var xi = d.X[i]` inside a loop: fine when this is the name's only assignment (it then always means "element i
        // as of here"; the index is frozen in a block-scoped temporary, so a use outside the loop does not compile instead of misreading)
        if (!(this.assignCount[name] === 1 && (v.t === 'rec' || v.t === 'dataArr' || v.t === 'stateArr' || v.t === 'strv' || v.t === 'strlit')))
          this.fail('aliasing an array or object (' + name + ') inside a loop or an if is only supported for a row / record assigned once (var row = data[i])');
        if (v.t === 'dataArr' && !/^\d+$/.test(v.off)) v = { t: 'dataArr', id: v.id, off: this.temp_int(v.off), dims: v.dims };
        if (v.t === 'stateArr' && !/^\d+$/.test(v.base)) v = { t: 'stateArr', base: this.temp_int(v.base), dims: v.dims };
        this.flush(out, indent);
      }
      if (Object.prototype.hasOwnProperty.call(this.localTypes, name)) this.fail(name + ' holds a number elsewhere and an array/object here');
      this.aliases[name] = v;
      return;
    }
    if (Object.prototype.hasOwnProperty.call(this.aliases, name)) this.fail(name + ' holds an array/object elsewhere and a number here');
    if (op === '=' && v.t === 'num' && v.cst !== undefined && this.assignCount[name] === 1 && name !== this.acc) {
      this.aliases[name] = v;     // assigned once, to a constant: it IS that constant (const N = d.y.length)
      return;
    }
    if (this.acc && name === this.acc) {
      if (op === '+=') {
        this.flush(out, indent);
        out.push(indent + ((ctx.split || !this.split) ? '' : 'if (sub == 0) ') + 'v_' + name + ' += ' + this.asD(v) + ';');
        return;
      }
      // the initialisation `var lp = <const>`
      this.setLocal(name, num('', false));
      this.flush(out, indent);
      out.push(indent + 'v_' + name + ' = ' + (this.split ? '(sub == 0) ? ' + this.asD(v) + ' : 0.0' : this.asD(v)) + ';');
      return;
    }
    if (op === '=') {
      this.setLocal(name, v);
      this.flush(out, indent);
      const isInt = this.localTypes[name] === 'int';
      out.push(indent + 'v_' + name + ' = ' + (isInt ? this.asI(v) : this.asD(v)) + ';');
      return;
    }
    if (!Object.prototype.hasOwnProperty.call(this.localTypes, name)) this.fail(name + ' is used before it is assigned');
    const cur = num('v_' + name, this.localTypes[name] === 'int');
    const bop = op[0];
    let res;
    if (bop === '/') res = num('(' + this.asD(cur) + ' / ' + this.asD(v) + ')', false);
    else if (bop === '%') res = num('js_mod(' + this.asD(cur) + ', ' + this.asD(v) + ')', false);
    else if (cur.int && v.int) res = num('(' + cur.code + ' ' + bop + ' ' + v.code + ')', true);
    else res = num('(' + this.asD(cur) + ' ' + bop + ' ' + this.asD(v) + ')', false);
    this.setLocal(name, res);
    this.flush(out, indent);
    out.push(indent + 'v_' + name + ' = ' + (this.localTypes[name] === 'int' ? this.asI(res) : this.asD(res)) + ';');
    return;
  }
  if (target.k === 'Index' && target.obj.k === 'Id' && this.aliases[target.obj.name] && this.aliases[target.obj.name].t === 'localArr' && this.aliases[target.obj.name].name) {
    const arr = this.aliases[target.obj.name];
    const idx = this.expr(target.idx), v = this.expr(valueAst);
    if (v.t !== 'num' && v.t !== 'bool') this.fail('array elements must be numbers');
    if (idx.cst !== undefined && !(Number.isInteger(idx.cst) && idx.cst >= 0 && idx.cst < arr.elems.length)) this.fail('index ' + idx.cst + ' is outside ' + target.obj.name);
    // a write outside the array would grow it in JavaScript; here the array has the length it was built with, such a write is dropped
    // (never performed on memory).  Indices that provably stay inside are not checked.
    const rg = rangeOf(idx);
    const safe = idx.cst !== undefined || (idx.int && rg && rg[0] >= 0 && rg[1] < arr.elems.length);
    let ix, guard = '';
    if (safe) ix = this.asI(idx);
    else if (idx.int) { ix = this.temp_int(this.asI(idx)); guard = 'if ((unsigned)' + ix + ' < ' + arr.elems.length + 'u) '; }
    else { const dv = this.temp(this.asD(idx)); const c = '(' + dv + ' >= 0.0 && ' + dv + ' < ' + arr.elems.length + '.0 && ' + dv + ' == __builtin_trunc(' + dv + '))'; ix = this.temp_int('(' + c + ' ? (int)' + dv + ' : 0)'); guard = 'if ' + c + ' '; }
    this.flush(out, indent);
    const lhs = arr.name + '[' + ix + ']';
    out.push(indent + guard + lhs + ' = ' + (op === '=' ? this.asD(v) : '(' + lhs + ' ' + op[0] + ' ' + this.asD(v) + ')') + ';');
    return;
  }
  if (target.k === 'Member' && target.obj.k === 'Id' && this.isStateName(target.obj.name)) {
    const key = target.prop;
    if (Object.prototype.hasOwnProperty.call(this.layout, key)) this.fail('log_post assigns to the parameter state.' + key + ' (only derived quantities may be assigned)');
    if (this.loops.length) this.fail('the derived quantity state.' + key + ' is assigned inside a loop');
    const v = this.expr(valueAst);
    if (v.t !== 'num' && v.t !== 'bool') this.fail('the derived quantity state.' + key + ' must be a number (array-valued derived quantities are not supported)');
    if (this.derived.indexOf(key) < 0) { if (op !== '=') this.fail('state.' + key + ' is updated before it is assigned'); this.derived.push(key); }
    this.flush(out, indent);
    const rhs = op === '=' ? this.asD(v) : '(dq_' + key + ' ' + op[0] + ' ' + this.asD(v) + ')';
    out.push(indent + 'dq_' + key + ' = ' + rhs + ';');
    return;
  }
  this.fail('assignment to this kind of target is not supported (only local variables and derived quantities state.key)');
};

Translator.prototype.canonicalLoop = function (s) {    // for (i = A; i < B; i++) with B fixed during the loop
  if (!s.init || !s.test || !s.update) return null;
  let name, startAst;
  if (s.init.k === 'VarDecl' && s.init.decls.length === 1 && s.init.decls[0].init) { name = s.init.decls[0].name; startAst = s.init.decls[0].init; }
  else if (s.init.k === 'ExprStmt' && s.init.expr.k === 'Assign' && s.init.expr.op === '=' && s.init.expr.target.k === 'Id') { name = s.init.expr.target.name; startAst = s.init.expr.value; }
  else return null;
  const t = s.test;
  if (t.k !== 'Binary' || (t.op !== '<' && t.op !== '<=') || t.l.k !== 'Id' || t.l.name !== name) return null;
  const u = s.update;
  const inc = (u.k === 'Update' && u.op === '++' && u.target.k === 'Id' && u.target.name === name) ||
              (u.k === 'Assign' && u.op === '+=' && u.target.k === 'Id' && u.target.name === name && u.value.k === 'Num' && u.value.v === 1);
  if (!inc) return null;
  const bodyAssigned = assignedNames(s.body);
  if (bodyAssigned.has(name)) return null;
  for (const nm of idsOf(t.r)) if (bodyAssigned.has(nm) || nm === name) return null;
  return { name, startAst, boundAst: t.r, le: t.op === '<=' };
};

// can the iterations of this top-level loop be dealt to the lanes of a chain?
Translator.prototype.splittable = function (s, canon) {
  if (!this.split || !canon) return false;
  if (containsKind(s.body, 'Return') || containsKind(s.body, 'Break')) return false;   // would have to stop the other lanes too
  let ok = true;
  const written = assignedNames(s.body);
  for (const a of this.accSet) written.delete(a);
  // every other variable written in the body must be private to one iteration: definitely assigned
  // before it is read in every iteration, and not referenced outside loops
  if (!definitelyAssigned(s.body.k === 'Block' ? s.body.body : [s.body], written, new Set(), this.accSet)) ok = false;
  for (const nm of written) if (this.topLevelRefs.has(nm)) ok = false;
  if (this.topLevelRefs.has(canon.name)) ok = false;
  // ... nor inside any LATER loop, unless that loop assigns them itself before it reads them (`var` is function-scoped: a temporary
  // of this loop read by a later loop would carry the value of the LAST iteration, which only one lane has)
  for (const other of this.topLoops.slice(this.topLoops.indexOf(s) + 1)) {      // top-level loops run once each, in source order
    const refs = idsOf(other);
    for (const nm of Array.from(written).concat([canon.name])) {
      if (!refs.has(nm)) continue;
      const initAssigns = other.init && assignedNames(other.init).has(nm);
      const header = new Set();
      if (!initAssigns) for (const part of [other.init, other.test, other.update]) if (part) idsOf(part).forEach((n) => header.add(n));
      if (header.has(nm)) { ok = false; continue; }
      if (initAssigns) continue;       // the other loop's own counter: assigned by its header before anything reads it
      if (!definitelyAssigned(other.body.k === 'Block' ? other.body.body : [other.body], new Set([nm]), new Set(), this.accSet)) ok = false;
    }
  }
  walk(s.body, (x) => {
    if (x.k === 'Assign' && x.target.k === 'Id' && this.accSet.has(x.target.name) && x.op !== '+=' && !(this.linear && x.op === '-=')) ok = false;
    if (x.k === 'Assign' && x.target.k !== 'Id') ok = false;   // derived quantities inside a loop
  });
  return ok;
};

Translator.prototype.stmt = function (s, out, indent, ctx) {
  switch (s.k) {
    case 'Empty': return;
    case 'Block': for (const x of s.body) this.stmt(x, out, indent, ctx); return;
    case 'VarDecl':
      for (const d of s.decls) {
        if (d.init) this.assign({ k: 'Id', name: d.name }, '=', d.init, out, indent, ctx);
        else if (!Object.prototype.hasOwnProperty.call(this.localTypes, d.name) && !Object.prototype.hasOwnProperty.call(this.aliases, d.name)) this.declaredOnly.add(d.name);
      }
      return;
    case 'ExprStmt': {
      const e = s.expr;
      if (e.k === 'Assign') return this.assign(e.target, e.op, e.value, out, indent, ctx);
      if (e.k === 'Update') return this.assign(e.target, e.op === '++' ? '+=' : '-=', { k: 'Num', v: 1 }, out, indent, ctx);
      if (e.k === 'Seq') { this.stmt({ k: 'ExprStmt', expr: e.l }, out, indent, ctx); this.stmt({ k: 'ExprStmt', expr: e.r }, out, indent, ctx); return; }
      this.fail('an expression statement that is not an assignment has no effect');
    }  // eslint-disable-line no-fallthrough
    case 'If': {
      const t = this.cond(s.test);
      this.flush(out, indent);
      if (t.cst !== undefined) { if (t.cst) this.stmt(s.cons, out, indent, ctx); else if (s.alt) this.stmt(s.alt, out, indent, ctx); return; }
      out.push(indent + 'if (' + t.code + ') {');
      this.condDepth = (this.condDepth || 0) + 1;
      this.stmt(s.cons, out, indent + '  ', ctx);
      if (s.alt) { out.push(indent + '} else {'); this.stmt(s.alt, out, indent + '  ', ctx); }
      this.condDepth--;
      out.push(indent + '}');
      return;
    }
    case 'Return': {
      if (!s.arg) this.fail('log_post returns nothing');
      if (this.acc && s.arg.k === 'Id' && s.arg.name === this.acc) {   // `return lp` (the last statement)
        out.push(indent + this.deriveStore());
        out.push(indent + 'return v_' + this.acc + ';');
        return;
      }
      if (ctx.split) this.fail('return inside a lane-split loop');   // excluded by splittable()
      if (this.linear) {
        const lc = this.linearCode(s.arg);
        this.flush(out, indent);
        out.push(indent + this.deriveStore());
        out.push(indent + 'return ' + (lc.hasA ? lc.code : '(sub == 0) ? ' + lc.code + ' : 0.0') + ';');
        return;
      }
      const v = this.expr(s.arg);
      this.flush(out, indent);
      out.push(indent + this.deriveStore());
      out.push(indent + 'return ' + (this.split ? '(sub == 0) ? ' + this.asD(v) + ' : 0.0' : this.asD(v)) + ';');
      return;
    }
    case 'For': return this.forLoop(s, out, indent, ctx);
    case 'Break': case 'Continue': {
      if (!this.loopLabels.length) this.fail("'" + (s.k === 'Break' ? 'break' : 'continue') + "' outside a loop");
      const lab = this.loopLabels[this.loopLabels.length - 1];
      this.flush(out, indent);
      // real C++ loops take the keyword; the while-form (init; test; body; update) needs the jump past / to its update
      out.push(indent + (lab.native ? (s.k === 'Break' ? 'break;' : 'continue;') : 'goto ' + (s.k === 'Break' ? 'brk_' : 'cont_') + lab.id + ';'));
      if (!lab.native) lab[s.k === 'Break' ? 'usedBreak' : 'usedContinue'] = true;
      return;
    }
  }
  this.fail('unsupported statement (' + s.k + ')');
};

Translator.prototype.deriveStore = function () {
  if (this.isHelper) return '';
  if (!this.derivedFinal || this.derivedFinal.length === 0) return 'if constexpr (DERIVE) { (void)dv; }';
  return 'if constexpr (DERIVE) { ' + this.derivedFinal.map((nm, q) => 'dv[' + q + '] = dq_' + nm + ';').join(' ') + ' }';
};

Translator.prototype.forLoop = function (s, out, indent, ctx) {
  const canon = this.canonicalLoop(s);
  const split = !ctx.inLoop && this.splittable(s, canon);
  if (split) this.nSplit++;
  const assigned = assignedNames(s);
  const L = { assigned, preamble: [] };
  const inner = [];
  const ind2 = indent + '    ';
  if (canon) {
    const startV = this.expr(canon.startAst), boundV = this.expr(canon.boundAst);
    this.flush(out, indent);
    this.setLocal(canon.name, startV);
    const isInt = this.localTypes[canon.name] === 'int';
    this.counterRange = this.counterRange || {};
    const hadRange = this.counterRange[canon.name];
    if (isInt && Number.isInteger(startV.cst) && Number.isInteger(boundV.cst)) this.counterRange[canon.name] = [startV.cst, canon.le ? boundV.cst : boundV.cst - 1];
    else delete this.counterRange[canon.name];
    L.restoreRange = () => { if (hadRange) this.counterRange[canon.name] = hadRange; else delete this.counterRange[canon.name]; };
    this.loops.push(L);
    const body = s.body.k === 'Block' ? s.body.body : [s.body];
    const bctx = { inLoop: true, split: split || ctx.split };
    // body of a lane-split loop that ends in `acc += term` (and writes acc nowhere else): U iterations are evaluated
    // independently (their temporaries are private, proved by splittable()), then the U terms are added in order
    const lastSt = body[body.length - 1];
    const endsInAcc = lastSt && lastSt.k === 'ExprStmt' && lastSt.expr.k === 'Assign' && lastSt.expr.op === '+=' &&
                      lastSt.expr.target.k === 'Id' && this.accSet.has(lastSt.expr.target.name);
    const loopAcc = endsInAcc ? lastSt.expr.target.name : this.acc;
    let accElsewhere = false;
    for (const st of body.slice(0, -1)) walk(st, (x) => { if ((x.k === 'Assign' || x.k === 'Update') && x.target.k === 'Id' && this.accSet.has(x.target.name)) accElsewhere = true; });
    const single = split && isInt && boundV.int && endsInAcc && !accElsewhere && !containsKind(s.body, 'Return') &&
                   !containsKind(s.body, 'Continue') && !containsKind(s.body, 'Break');
    if (this.loopInfo) this.loopInfo.push({ name: canon.name, single: !!single, start: startV.cst, bound: boundV.cst, le: !!canon.le });
    if (single) {
      const pre = [];
      const heavyBefore = this.heavyLoop;
      for (const st of body.slice(0, -1)) this.stmt(st, pre, '', bctx);
      const term = this.expr(lastSt.expr.value);
      const pend = this.pending; this.pending = [];
      this.loops.pop();
      L.restoreRange();
      const simple = body.length === 1;
      // plain arithmetic bodies keep 8 terms in flight; bodies with exp/log/ld.* calls 4 (register pressure)
      const U = this.opts.unroll || ((this.heavyLoop && !simple) || (this.heavyLoop && !heavyBefore && pend.length > 4) ? 4 : 8);
      const acc = 'v_' + loopAcc, iv = 'v_' + canon.name;
      const bodyText = pre.map((ln) => ln.trim()).concat(pend).join(' ');
      const loop = [];
      loop.push('  int it_ = 0;');
      loop.push('  for (; it_ + ' + U + ' <= n_; it_ += ' + U + ') {');
      loop.push('    double tb_[' + U + '];');
      loop.push('#pragma unroll');
      // a softplus in the term (csrc/amwg_math.h log1p_exp_v8): the U terms are formed by its branch-free form, one after the other in a single basic
      // block, and a lane that met an argument it does not cover (flagged, a handful per million) forms its U terms again through the full functions
      const termText = this.asD(term), softplus = /\blog1p_exp_v8\(/.test(bodyText + termText) && !this.opts.no_open_softplus;
      if (softplus) {
        const open = (t) => t.replace(/\blog1p_exp_v8\(/g, 'log1p_exp_v8_open(rr_, '), cold = (t) => t.replace(/\blog1p_exp_v8\(/g, 'log1p_exp_cold(');
        loop.pop();
        loop.push('    bool rr_ = false;');
        loop.push('#pragma unroll');
        loop.push('    for (int u_ = 0; u_ < ' + U + '; ++u_) { const int ' + iv + ' = i0_ + (it_ + u_) * G; ' + open(bodyText) + ' tb_[u_] = ' + open(termText) + '; }');
        loop.push('    if (rr_) {');
        loop.push('#pragma unroll');
        loop.push('      for (int u_ = 0; u_ < ' + U + '; ++u_) { const int ' + iv + ' = i0_ + (it_ + u_) * G; ' + cold(bodyText) + ' tb_[u_] = ' + cold(termText) + '; }');
        loop.push('    }');
      } else
      loop.push('    for (int u_ = 0; u_ < ' + U + '; ++u_) { const int ' + iv + ' = i0_ + (it_ + u_) * G; ' + bodyText + ' tb_[u_] = ' + this.asD(term) + '; }');
      loop.push('#pragma unroll');
      loop.push('    for (int u_ = 0; u_ < ' + U + '; ++u_) ' + acc + ' += tb_[u_];');
      loop.push('  }');
      loop.push('  for (; it_ < n_; ++it_) { const int ' + iv + ' = i0_ + it_ * G; ' + bodyText + ' ' + acc + ' += ' + this.asD(term) + '; }');
      const head = ['  const int i0_ = ' + this.asI(startV) + ' + sub, n_ = (' + boundV.code + (canon.le ? ' + 1' : '') + ' - i0_ + G - 1) / G;'];
      // `for (i = 0; i < x.length; i++) lp += ld.bern(x[i], p)` over a 0/1 array: with one lane per chain the sequential sum has
      // two distinct addends and is fast-forwarded exactly (csrc/amwg_twoval.h); with G > 1 lanes the ordinary split loop runs
      const mb = simple && pend.length === 0 && /^ld_bern_inv01\(A(\d+)\[v_(\w+)\], (k\d+)\)$/.exec(term.code);
      if (mb && mb[2] === canon.name && startV.cst === 0 && !canon.le && boundV.cst === this.arrays[Number(mb[1])].flat.length && !this.opts.no_fast_forward) {
        const tabId = this.twoValuedTables(Number(mb[1]));
        s.fastForwardOneLane = true;     // workEstimate(true) prices this loop as ~log2(n) binades
        this.oneLaneWork = 1;
        out.push(indent + '{');
        for (const ln of L.preamble) out.push(indent + '  ' + ln);
        out.push(indent + '  if constexpr (G == 1) {');
        out.push(indent + '    ' + acc + ' = bern_loop_one_lane(' + acc + ', ' + mb[3] + ', A' + tabId + ', ' + boundV.cst + ');');
        out.push(indent + '  } else {');
        for (const ln of head.concat(loop)) out.push(indent + '  ' + ln);
        out.push(indent + '  }');
        out.push(indent + '}');
        return;
      }
      // `for (i = 0; i < y.length; i++) lp += ld.pois(y[i], rate)` / `ld.binom(y[i], size, prob)` with loop-invariant rate / size / prob over a
      // whole array of at most 16 distinct values: the term is a function of y[i] alone, so with one lane per chain the sequential sum has
      // K distinct addends and is fast-forwarded exactly (csrc/amwg_kval.h); with G > 1 lanes the ordinary split loop runs
      // (any density whose only per-observation argument is y[i] qualifies; ld.pois arrives with lfactorial(y[i]) precomputed as a second array)
      let mk = null;
      if (simple && pend.length === 0 && !this.opts.no_fast_forward) {
        const m1 = /^(ld_pois_pre)\(((?:\(double\))?)A(\d+)\[v_(\w+)\], (.+), A(\d+)\[v_\4\]\)$/.exec(term.code);
        const m2 = m1 ? null : /^(ld_\w+)\(((?:\(double\))?)A(\d+)\[v_(\w+)\], (.+)\)$/.exec(term.code);
        const m = m1 || m2;
        if (m && m[4] === canon.name && m[5].indexOf('v_' + canon.name) < 0 && !/A\d+\[/.test(m[5]))
          mk = { fn: m[1], cast: m[2], arr: Number(m[3]), rest: m[5], aux: m1 ? Number(m1[6]) : -1 };
      }
      if (mk && startV.cst === 0 && !canon.le && boundV.cst === this.arrays[mk.arr].flat.length && this.arrays[mk.arr].type !== 0) {      // (a small-integer array: stored as u8 / i32)
        const kv = this.kValuedTables(mk.arr);
        if (kv) {
          s.fastForwardOneLane = true;
          this.oneLaneWork = 1;
          out.push(indent + '{');
          for (const ln of L.preamble) out.push(indent + '  ' + ln);
          out.push(indent + '  if constexpr (G == 1) {');
          out.push(indent + '    const double c_[' + kv.K + '] = {' + kv.first.map((i0) => mk.fn + '(' + mk.cast + 'A' + mk.arr + '[' + i0 + '], ' + mk.rest + (mk.aux >= 0 ? ', A' + mk.aux + '[' + i0 + ']' : '') + ')').join(', ') + '};');
          out.push(indent + '    ' + acc + ' = kval_loop_one_lane<' + kv.K + '>(' + acc + ', c_, A' + kv.tab + ', A' + kv.idx + ', ' + boundV.cst + ');');
          out.push(indent + '  } else {');
          for (const ln of head.concat(loop)) out.push(indent + '  ' + ln);
          out.push(indent + '  }');
          out.push(indent + '}');
          return;
        }
      }
      // `for (i = 0; i < x.length; i++) lp += ld.norm(x[i], mean, sd)` over a whole f64 data array, mean and sd loop-invariant: the
      // hand-scheduled pass the built-in Normal family runs (csrc/amwg_pass.h via norm_data_loop, amwg_user.h) -- staged LDS reads
      // with G lanes, scalar loads with one lane per chain; same operations in the same order as the generic loop below
      const mn = simple && pend.length === 0 && /^NORMCALL\((?:\(double\))?A(\d+)\[v_(\w+)\], (.+), (k\d+) RANGEARGS\)$/.exec(term.code);
      if (mn && mn[2] === canon.name && startV.cst === 0 && !canon.le &&
          boundV.cst === this.arrays[Number(mn[1])].flat.length && mn[3].indexOf('v_' + canon.name) < 0 && mn[3].indexOf('NORMCALL') < 0 && !this.opts.no_staged_norm) {
        const arr = this.arrays[Number(mn[1])];
        let mid = true;
        for (let i = 0; i < arr.flat.length && mid; i++) { const v = Math.abs(arr.flat[i]); mid = v === 0 || (v >= Math.pow(2, -200) && v <= Math.pow(2, 200)); }
        if (arr.ctype === 'double') this.uniformNormLoops = (this.uniformNormLoops || 0) + 1;      // scalar-load pass with one lane per chain
        else this.otherSplitLoops = (this.otherSplitLoops || 0) + 1;
        // CERTIFIED TAIL candidate (csrc/amwg_user.h norm_tail_approx): this loop at the top level of the closure, adding to the closure's one accumulator, f64 observations,
        // mean and sd expressions of the state alone.  run() decides (it must be the LAST statement: tailPlan).
        // (the loop's hoisted preamble: `const double kN = <expression of the state>;` lines -- sub-expressions of sd -- and the NormInv line last; they travel with sd)
        const nPre = L.preamble.length;
        const sdT = nPre >= 1 && /^const NormInv (k\d+) = norm_inv\((.+)\);(?: KFASTCHECK\(\1\))?$/.exec(L.preamble[nPre - 1].trim());
        const stateOnly = (e, okNames) => !(e.match(/\b(v_\w+|A\d+|t\d+|k\d+|tb_\w*|it_\w*|sub|dq_\w+)\b/g) || []).some((w) => !(okNames || []).includes(w));
        const sdLets = [];
        let sdOk = !!sdT;
        for (let q = 0; sdOk && q < nPre - 1; q++) {
          const m2 = /^const double (k\d+) = (.+);$/.exec(L.preamble[q].trim());
          if (m2 && stateOnly(m2[2], sdLets.map((z) => z[0]))) sdLets.push([m2[1], m2[2]]); else sdOk = false;
        }
        const tailCand = !this.isHelper && !this.opts.no_cert_tail && !this.linear && this.acc && loopAcc === this.acc && indent === '    ' && !this.condDepth &&
                         arr.ctype === 'double' && boundV.cst >= 1 && sdOk && sdT[1] === mn[4] && stateOnly(sdT[2], sdLets.map((z) => z[0])) && stateOnly(mn[3]);
        const sdText = tailCand ? (sdLets.length ? '[&] { ' + sdLets.map((z) => 'const double ' + z[0] + ' = ' + z[1] + '; ').join('') + 'return ' + sdT[2] + '; }()' : sdT[2]) : '';
        if (tailCand) out.push(indent + '//@TAIL x=A' + mn[1] + ' n=' + boundV.cst + ' acc=' + loopAcc + ' mean=' + mn[3] + ' @sd=' + sdText);
        out.push(indent + '{');
        for (const ln of renderNorm(L.preamble.map((q) => '  ' + q), 'inv')) out.push(indent + ln);
        out.push(indent + '  ' + acc + ' = norm_data_loop<G>(A' + mn[1] + ', static_cast<const ' + arr.ctype + ' *>(user_arr<' + mn[1] + '>(d)), ' + boundV.cst + ', ' + mn[3] + ', ' + mn[4] + ', ' + (mid ? 'true' : 'false') + ', sub, ' + acc + ');');
        out.push(indent + '}');
        if (tailCand) out.push(indent + '//@TAIL_END');
        return;
      }
      // ... and with a GATHERED mean, `lp += ld.norm(y[i], state.theta[g[i]], sd)`: g a byte-typed data array whose values index a
      // parameter vector (provably in range: no guard in the generated read).  norm_data_loop_gather (amwg_user.h) runs the staged pass;
      // for lane counts at which the labels repeat with the lane stride (g[i] == g[i % G]) a lane reads its one mean once.
      const mg = simple && pend.length === 0 && /^NORMCALL\((?:\(double\))?A(\d+)\[v_(\w+)\], S\((?:(\d+) \+ )?\(int\)A(\d+)\[v_(\w+)\]\), (k\d+) RANGEARGS\)$/.exec(term.code);
      if (mg && mg[2] === canon.name && mg[5] === canon.name && startV.cst === 0 && !canon.le &&
          this.arrays[Number(mg[4])].ctype === 'uint8_t' && boundV.cst === this.arrays[Number(mg[1])].flat.length &&
          boundV.cst <= this.arrays[Number(mg[4])].flat.length && !this.opts.no_staged_norm) {
        const arr = this.arrays[Number(mg[1])], gl = this.arrays[Number(mg[4])].flat, n = boundV.cst;
        let mid = true, ng = 0, mask = 0;
        for (let i = 0; i < n && mid; i++) { const v = Math.abs(arr.flat[i]); mid = v === 0 || (v >= Math.pow(2, -200) && v <= Math.pow(2, 200)); }
        for (let i = 0; i < n; i++) ng = Math.max(ng, gl[i] + 1);
        for (let j = 0; j <= 10; j++) { const Gj = 1 << j; let per = n > 0; for (let i = Gj; i < n && per; i++) per = gl[i] === gl[i % Gj]; if (per) mask |= 1 << j; }
        // ROW PLAN candidate (csrc/amwg_rows.h): this loop at the top level of the closure, adding to the closure's one accumulator, f64 observations,
        // labels that repeat with a stride of 64, an sd that is an expression of the state alone.  run() decides (it must be the LAST statement).
        const sdLine = L.preamble.length === 1 && /^const NormInv (k\d+) = norm_inv\((.+)\);(?: KFASTCHECK\(\1\))?$/.exec(L.preamble[0].trim());
        const rowCand = !this.isHelper && !this.opts.no_row_plan && !this.linear && this.acc && loopAcc === this.acc && indent === '    ' && !this.condDepth &&
                        arr.ctype === 'double' && n >= 64 && ng >= 1 && ng <= 64 && ((mask >> 6) & 1) && sdLine && sdLine[1] === mg[6] &&
                        !/\b(v_\w+|A\d+|t\d+|k\d+|tb_|it_|sub|dq_\w+)\b/.test(sdLine[2]);
        if (rowCand) out.push(indent + '//@ROWS y=A' + mg[1] + ' labels=A' + mg[4] + ' base=' + (mg[3] || '0') + ' ng=' + ng + ' n=' + n + ' mid=' + (mid ? 1 : 0) + ' acc=' + loopAcc + ' sd=' + sdLine[2]);
        out.push(indent + '{');
        for (const ln of renderNorm(L.preamble.map((q) => '  ' + q), 'inv')) out.push(indent + ln);
        out.push(indent + '  ' + acc + ' = norm_data_loop_gather<G, ((' + mask + 'u >> __builtin_ctz((unsigned)G)) & 1u) != 0u>(A' + mg[1] + ', A' + mg[4] + ', S, ' + (mg[3] || '0') + ', ' + ng + ', ' + n + ', ' + mg[6] + ', ' + (mid ? 'true' : 'false') + ', sub, ' + acc + ');');
        out.push(indent + '}');
        if (rowCand) out.push(indent + '//@ROWS_END');
        this.otherSplitLoops = (this.otherSplitLoops || 0) + 1;      // (one lane per chain gains nothing here: the data stays in LDS)
        return;
      }
      // `for (i = 0; i < y.length; i++) { <eta from the state and row i>; lp += ld.pois(y[i], Math.exp(eta)); }` over whole arrays of counts: CERTIFIED POISSON TAIL
      // candidate (csrc/amwg_ptail.h).  run() decides (it must be the LAST statement: poisTailPlan); the loop itself is emitted as any other lane-split loop.
      const mp = /^ld_pois_pre_exp\(((?:\(double\))?)A(\d+)\[v_(\w+)\], (.+), A(\d+)\[v_\3\]\)$/.exec(term.code);
      let ptailCand = false;
      if (mp && mp[3] === canon.name && startV.cst === 0 && !canon.le && L.preamble.length === 0 && !this.isHelper && !this.opts.no_cert_tail && !this.linear && this.acc &&
          loopAcc === this.acc && indent === '    ' && !this.condDepth) {
        const ya = this.arrays[Number(mp[2])], la = this.arrays[Number(mp[5])], text = bodyText + ' ' + mp[4];
        let ok = ya.type !== 0 && boundV.cst === ya.flat.length && boundV.cst === la.flat.length && boundV.cst >= 64 && !/\b(sub|G|dq_\w+|dv|return|goto|tb_\w*|it_\w*|u_|rr_)\b/.test(text);
        for (let i = 0; ok && i < ya.flat.length; i++) ok = ya.flat[i] >= 0 && Number.isFinite(la.flat[i]);      // (a negative count: the reference's term is -inf)
        // are the state's entries the statements read the same for every observation?  Every S(index): a constant, or constant + the counter of an inner loop with
        // constant bounds (and nothing else writes that counter).  Then the pass keeps the four chains' entries in scalar registers (UniformState).
        let uniform = ok;
        if (ok) {
          const counters = new Set();
          let rest = text.replace(/for \((v_\w+) = (\d+); \1 < (\d+); \1 \+= 1\)/g, (q, nm) => { counters.add(nm); return 'for ()'; });
          for (const nm of counters) if (new RegExp('\\b' + nm.replace(/[$]/g, '\\$') + '\\s*(?:[-+*/%]?=[^=]|\\+\\+|--)').test(rest) || nm === 'v_' + canon.name) uniform = false;
          for (let at = text.indexOf('S('); uniform && at >= 0; at = text.indexOf('S(', at + 1)) {
            if (at > 0 && /[\w.]/.test(text[at - 1])) continue;      // (an identifier that merely ends in S)
            let depth = 0, end = at + 1;
            for (; end < text.length; end++) { if (text[end] === '(') depth++; else if (text[end] === ')' && --depth === 0) break; }
            const arg = text.slice(at + 2, end), ma = /^(?:\d+|(?:\d+ \+ )?(v_\w+))$/.exec(arg);
            if (!ma || (ma[1] && !counters.has(ma[1]))) uniform = false;
          }
          if (/\bS\b(?!\()/.test(text)) ok = false;      // (the state handed on as a whole)
        }
        // ROW CACHE: does every read of a data array in the statements address observation i's own row -- `A[i]`, `A[i * K + c]`, `A[i * K + counter]` (c, the counter's
        // range inside [0, K))?  Then the pass loads a row into registers a round AHEAD of its use (ptail_load / ptail_eta_row: the statements with the reads replaced by
        // the row's entries) instead of waiting for every row where its first product needs it.
        let rowInfo = null;
        if (ok && uniform && !this.opts.no_tail_rows) {
          const iv = 'v_' + canon.name, ivE = iv.replace(/[$]/g, '\\$'), cb = {};
          text.replace(/for \((v_\w+) = (\d+); \1 < (\d+); \1 \+= 1\)/g, (q, nm, lo, hi) => { cb[nm] = cb[nm] ? [Math.min(cb[nm][0], +lo), Math.max(cb[nm][1], +hi)] : [+lo, +hi]; return q; });
          const per = {};
          per[Number(mp[2])] = { stride: 1, lo: 0, hi: 1 };      // (the count itself)
          const scan = (t) => {      // -> [{at, end, j, rel}] or null
            const reads = [], re = /\bA(\d+)\[/g;
            let m;
            while ((m = re.exec(t))) {
              let depth = 0, end = m.index + m[0].length - 1;
              for (; end < t.length; end++) { if (t[end] === '[') depth++; else if (t[end] === ']' && --depth === 0) break; }
              const idx = t.slice(m.index + m[0].length, end), j = Number(m[1]);
              let mm, stride, lo, hi, rel = '0';
              if (idx === iv) { stride = 1; lo = 0; hi = 1; }
              else if ((mm = new RegExp('^\\(' + ivE + ' \\* (\\d+)\\)$').exec(idx))) { stride = +mm[1]; lo = 0; hi = 1; }
              else if ((mm = new RegExp('^\\(\\(' + ivE + ' \\* (\\d+)\\) \\+ (\\d+|v_\\w+)\\)$').exec(idx))) {
                stride = +mm[1]; rel = mm[2];
                if (/^\d+$/.test(rel)) { lo = +rel; hi = lo + 1; } else if (cb[rel]) { lo = cb[rel][0]; hi = cb[rel][1]; } else return null;
              } else return null;
              if (lo < 0 || hi > stride || (per[j] && per[j].stride !== stride) || stride * boundV.cst > this.arrays[j].flat.length) return null;
              per[j] = per[j] ? { stride, lo: Math.min(per[j].lo, lo), hi: Math.max(per[j].hi, hi) } : { stride, lo, hi };
              reads.push({ at: m.index, end, j, rel });
            }
            return reads;
          };
          const rb = scan(bodyText), re2 = scan(mp[4]);
          let words = 0;
          for (const j of Object.keys(per)) words += (per[j].hi - per[j].lo) * (this.arrays[j].ctype === 'double' ? 2 : 1);
          if (rb && re2 && words <= 40) {
            const rewrite = (t, reads) => { let o = t; for (let q = reads.length - 1; q >= 0; q--) { const r = reads[q]; o = o.slice(0, r.at) + 'R.a' + r.j + '[(' + r.rel + ') - ' + per[r.j].lo + ']' + o.slice(r.end + 1); } return o; };
            rowInfo = { per, body: rewrite(bodyText, rb), eta: rewrite(mp[4], re2) };
          }
        }
        // LINEAR PREDICTOR: is eta nothing but a sum of products (row entry) x (state entry), state entries and literals -- `eta = 0; for (k) eta += X[i K + k] * b[k];
        // if (...) eta += b[7]`?  Then the pass forms it by fused steps (one rounding per product instead of two) and bounds |eta| AND the distance between the two
        // etas by H = sum |b_k| max_i |x_ik| + ..., once per pass, instead of taking max |eta_i| over the rows (csrc/amwg_ptail.h, kTailLinear).
        let linInfo = null;
        if (rowInfo && /^v_\w+$/.test(rowInfo.eta) && !this.opts.no_tail_linear) {
          const E = rowInfo.eta, Ee = E.replace(/[$]/g, '\\$'), T = rowInfo.body;
          const loops = [];      // {name, lo, hi, from, to}: the constant-bound loops of the statements and the extent of their bodies in T
          const reFor = /for \((v_\w+) = (\d+); \1 < (\d+); \1 \+= 1\) \{/g;
          let mf, good = true;
          while ((mf = reFor.exec(T))) {
            let depth = 0, end = mf.index + mf[0].length - 1;
            for (; end < T.length; end++) { if (T[end] === '{') depth++; else if (T[end] === '}' && --depth === 0) break; }
            loops.push({ name: mf[1], lo: +mf[2], hi: +mf[3], from: mf.index, to: end });
          }
          const colmax = (j, stride, o) => { let m = 0; const f = this.arrays[j].flat; for (let i = 0; i < boundV.cst; i++) m = Math.max(m, Math.abs(f[i * stride + o])); return m; };
          const terms = [];      // [text of one summand of H]
          let refRound = 0, fusedRound = 0;
          const reAsg = new RegExp(Ee + ' = ([^;]+);', 'g');
          let fused = '', last = 0, ma;
          while (good && (ma = reAsg.exec(T))) {
            const rhs = ma[1], at = ma.index;
            const inLoops = loops.filter((q) => at > q.from && at < q.to);
            const tripOthers = (used) => inLoops.filter((q) => q.name !== used).reduce((t, q) => t * (q.hi - q.lo), 1);
            const trips = inLoops.reduce((t, q) => t * (q.hi - q.lo), 1);      // how often the statement runs per observation (at most)
            let mm, out = null;
            if (/^-?(?:\d+\.?\d*(?:e[-+]?\d+)?|0x[\da-f.]+p[-+]?\d+)$/i.test(rhs)) {      // eta = literal
              if (inLoops.length) good = false;
              const v = Math.abs(parseLiteral(rhs));
              if (!(v >= 0)) good = false; else if (v > 0) terms.push(hexFloat(v));
              out = ma[0];
            } else if ((mm = new RegExp('^\\(' + Ee + ' ([-+]) \\((R\\.a(\\d+)\\[\\((\\d+|v_\\w+)\\) - (\\d+)\\]) \\* (S\\(((?:\\d+ \\+ )?)(\\d+|v_\\w+)\\))\\)\\)$').exec(rhs)) ||
                       (mm = new RegExp('^\\(' + Ee + ' ([-+]) \\((S\\(((?:\\d+ \\+ )?)(\\d+|v_\\w+)\\)) \\* (R\\.a(\\d+)\\[\\((\\d+|v_\\w+)\\) - (\\d+)\\])\\)\\)$').exec(rhs))) {
              // (two orders of the factors: normalise to data x state)
              const dataFirst = mm[2].indexOf('R.') === 0;
              const sign = mm[1], rd = dataFirst ? mm[2] : mm[5], j = +(dataFirst ? mm[3] : mm[6]), rel = dataFirst ? mm[4] : mm[7], sv = dataFirst ? mm[6] : mm[2], sbase = (dataFirst ? mm[7] : mm[3]), sidx = dataFirst ? mm[8] : mm[4];
              const per = rowInfo.per[j], ctr = /^v_/.test(rel) ? rel : (/^v_/.test(sidx) ? sidx : null);
              if ((/^v_/.test(rel) && /^v_/.test(sidx) && rel !== sidx) || (ctr && !inLoops.some((q) => q.name === ctr))) good = false;
              else {
                const q = ctr ? inLoops.find((z) => z.name === ctr) : null, mult = tripOthers(ctr);
                const b0 = sbase ? parseInt(sbase, 10) : 0;
                for (let c = q ? q.lo : 0; c < (q ? q.hi : 1); c++) {
                  const o = /^v_/.test(rel) ? c : +rel, si = b0 + (/^v_/.test(sidx) ? c : +sidx);
                  if (o < per.lo || o >= per.hi || si < 0 || si >= this.P) { good = false; break; }
                  const cm = colmax(j, per.stride, o) * mult;
                  if (!Number.isFinite(cm)) { good = false; break; }
                  if (cm > 0) terms.push(hexFloat(cm) + ' * __builtin_fabs(S(' + si + '))');
                }
                refRound += 2 * trips; fusedRound += trips;
                out = E + ' = __builtin_fma(' + (sign === '-' ? '-' : '') + rd + ', ' + sv + ', ' + E + ');';
              }
            } else if ((mm = new RegExp('^\\(' + Ee + ' [-+] S\\(((?:\\d+ \\+ )?)(\\d+|v_\\w+)\\)\\)$').exec(rhs))) {      // eta +- a state entry
              const ctr = /^v_/.test(mm[2]) ? mm[2] : null;
              if (ctr && !inLoops.some((q) => q.name === ctr)) good = false;
              else {
                const q = ctr ? inLoops.find((z) => z.name === ctr) : null, mult = tripOthers(ctr), b0 = mm[1] ? parseInt(mm[1], 10) : 0;
                for (let c = q ? q.lo : 0; c < (q ? q.hi : 1); c++) {
                  const si = b0 + (ctr ? c : +mm[2]);
                  if (si < 0 || si >= this.P) { good = false; break; }
                  terms.push((mult > 1 ? mult + '.0 * ' : '') + '__builtin_fabs(S(' + si + '))');
                }
                refRound += trips; fusedRound += trips;
                out = ma[0];
              }
            } else good = false;
            if (good) { fused += T.slice(last, at) + out; last = at + ma[0].length; }
          }
          if (good) {
            fused += T.slice(last);
            // eta appears nowhere but in those assignments (a condition on eta, eta handed to a function: not a linear predictor)
            const rest = T.replace(reAsg, ';');
            if (new RegExp('\\b' + Ee + '\\b').test(rest) || !terms.length || terms.length > 64 || refRound + fusedRound > 200) good = false;
          }
          if (good) linInfo = { body: fused, hlin: terms.join(' + '), roundings: refRound + fusedRound };
        }
        if (ok) {
          ptailCand = true;
          this.ptailInfo = { y: Number(mp[2]), lf: Number(mp[5]), cast: mp[1], n: boundV.cst, acc: loopAcc, i: canon.name, uniform, eta: mp[4], body: bodyText, rows: rowInfo, linear: linInfo };
          out.push(indent + '//@PTAIL');
        }
      }
      this.otherSplitLoops = (this.otherSplitLoops || 0) + 1;
      this.emitSplit(out, indent, L.preamble, head, loop, [loopAcc]);
      if (ptailCand) out.push(indent + '//@PTAIL_END');
      return;
    }
    this.loopLabels.push({ native: true });
    for (const x of body) this.stmt(x, inner, ind2, bctx);
    this.loopLabels.pop();
    this.loops.pop();
    L.restoreRange();
    out.push(indent + '{');
    for (const p of L.preamble) out.push(indent + '  ' + p);
    const v = 'v_' + canon.name;
    const cmp = canon.le ? ' <= ' : ' < ';
    const bound = isInt && boundV.int ? boundV.code : this.asD(boundV);
    const lhs = isInt && !boundV.int ? '(double)' + v : v;
    const start = isInt ? this.asI(startV) : this.asD(startV);
    if (split) {
      const loop = ['  for (' + v + ' = ' + start + ' + sub; ' + lhs + cmp + bound + '; ' + v + ' += G) {'];
      for (const ln of inner) loop.push(ln.slice(indent.length));
      loop.push('  }');
      out.pop();   // the '{' pushed above: emitSplit writes its own block
      L.preamble.forEach(() => out.pop());
      this.otherSplitLoops = (this.otherSplitLoops || 0) + 1;
      this.emitSplit(out, indent, L.preamble, [], loop, Array.from(this.accSet).filter((a) => assigned.has(a)));
      return;
    }
    out.push(indent + '  for (' + v + ' = ' + start + '; ' + lhs + cmp + bound + '; ' + v + ' += 1) {');
    for (const ln of inner) out.push(ln);
    out.push(indent + '  }');
    out.push(indent + '}');
    return;
  }
  // general loop: init; while (test) { body; update; }
  if (s.init) this.stmt(s.init, out, indent, ctx);
  this.loops.push(L);
  const bctx = { inLoop: true, split: ctx.split };
  this.noHoist = true;   // temporaries of the condition would have to be recomputed; keep it simple
  const t = s.test ? this.cond(s.test) : { code: 'true' };
  if (this.pending.length) this.fail('this loop condition is too complex (it needs temporaries)');
  this.noHoist = false;
  const lab = { native: false, id: this.labelSeq++ };
  this.loopLabels.push(lab);
  this.stmt(s.body, inner, ind2 + '  ', bctx);
  this.loopLabels.pop();
  const upd = [];
  if (s.update) this.stmt({ k: 'ExprStmt', expr: s.update }, upd, ind2, bctx);
  this.loops.pop();
  out.push(indent + '{');
  for (const p of L.preamble) out.push(indent + '  ' + p);
  out.push(indent + '  while (' + t.code + ') {');
  out.push(ind2 + '{');                         // own scope: `continue` jumps out of it, past no initialisation
  for (const ln of inner) out.push(ln);
  out.push(ind2 + '}');
  if (lab.usedContinue) out.push(ind2 + 'cont_' + lab.id + ': ;');
  for (const ln of upd) out.push(ln);
  out.push(indent + '  }');
  if (lab.usedBreak) out.push(indent + '  brk_' + lab.id + ': ;');
  out.push(indent + '}');
};

// A lane-split loop.  If it evaluates ld.norm with a hoisted sd, it is emitted twice: the fast form (4-operation
// quotient, exponent range of the numerators recorded) and, run only if a range precondition failed, the slow form
// (IEEE division) from the saved accumulator -- the two give the same bits, the second is the proof obligation.
Translator.prototype.emitSplit = function (out, indent, preamble, head, loop, accNames) {
  // accNames: the accumulators this loop adds to (saved before the fast pass, restored before the IEEE replay)
  const accs = (accNames && accNames.length ? accNames : [this.acc]).map((a) => 'v_' + a);
  const all = preamble.concat(head, loop);
  out.push(indent + '{');
  if (!hasNormCall(all)) {
    for (const ln of renderNorm(preamble.map((p) => '  ' + p).concat(head, loop), 'inv')) out.push(indent + ln);
    out.push(indent + '}');
    return;
  }
  accs.forEach((acc, k) => out.push(indent + '  const double acc_save' + (k ? k : '') + '_ = ' + acc + ';'));
  out.push(indent + '  uint32_t rlo_ = 0xffffffffu, rhi_ = 0u;');
  for (const ln of renderNorm(preamble.map((p) => '  ' + p), 'fast')) out.push(indent + ln);
  for (const ln of head) out.push(indent + ln);
  out.push(indent + '  {');
  for (const ln of renderNorm(loop, 'fast')) out.push(indent + '  ' + ln);
  out.push(indent + '  }');
  out.push(indent + '  if (!norm_range_ok(rlo_, rhi_)) {');
  accs.forEach((acc, k) => out.push(indent + '    ' + acc + ' = acc_save' + (k ? k : '') + '_;'));
  for (const ln of renderNorm(loop, 'slow')) out.push(indent + '  ' + ln);
  out.push(indent + '  }');
  out.push(indent + '}');
};

// The largest set of locals that (1) are written only by `=`, `+=`, `-=` (inside loops only `+=` / `-=` of accumulator-free terms),
// (2) are read only in linear positions (sums, differences, products with / quotients by accumulator-free factors, negation) of
// assignments to members of the set or of a return, and (3) actually accumulate (`+=`) or collect other members; empty unless a
// return depends on it.  Lanes then hold partial values whose sum over the lanes is the sequential value.
Translator.prototype.linearAccumulators = function (body) {
  const writes = {};                      // name -> [{op, value, depth}]
  const bad = new Set();
  const noteWrite = (name, op, value, depth) => { (writes[name] = writes[name] || []).push({ op, value, depth }); };
  const scanW = (node, depth) => {
    if (!node || typeof node !== 'object') return;
    if (Array.isArray(node)) { node.forEach((x) => scanW(x, depth)); return; }
    if (node.k === 'Func') return;
    if (node.k === 'VarDecl') node.decls.forEach((d) => { if (d.init) noteWrite(d.name, '=', d.init, depth); });
    if (node.k === 'Assign' && node.target.k === 'Id') noteWrite(node.target.name, node.op, node.value, depth);
    if (node.k === 'Update' && node.target.k === 'Id') bad.add(node.target.name);
    if (node.k === 'For') { scanW(node.init, depth); scanW(node.test, depth + 1); scanW(node.update, depth + 1); scanW(node.body, depth + 1); return; }
    for (const key of Object.keys(node)) if (key !== 'k') scanW(node[key], depth);
  };
  scanW(body, 0);
  let C = new Set(Object.keys(writes).filter((n) => !bad.has(n) && writes[n].every((w) => w.op === '=' || w.op === '+=' || w.op === '-=')));
  if (this.ast.params) for (const q of this.ast.params) C.delete(q);
  const members = (e) => { const o = []; for (const nm of idsOf(e)) if (C.has(nm)) o.push(nm); return o; };
  // linear(e): null if some member of C sits in a non-linear position of e, else whether e contains members at all
  const linear = (e) => {
    if (!e || !members(e).length) return false;
    if (e.k === 'Id') return true;
    if (e.k === 'Unary' && (e.op === '-' || e.op === '+')) return linear(e.arg);
    if (e.k === 'Binary' && (e.op === '+' || e.op === '-')) { const l = linear(e.l), r = linear(e.r); return (l === null || r === null) ? null : (l || r); }
    if (e.k === 'Binary' && e.op === '*') { const ml = members(e.l).length, mr = members(e.r).length; if (ml && mr) return null; return ml ? linear(e.l) : linear(e.r); }
    if (e.k === 'Binary' && e.op === '/') { if (members(e.r).length) return null; return linear(e.l); }
    return null;
  };
  for (let changed = true; changed;) {
    changed = false;
    const drop = (names) => { for (const nm of names) if (C.delete(nm)) changed = true; };
    // reads
    const visit = (node, depth) => {
      if (!node || typeof node !== 'object') return;
      if (Array.isArray(node)) { node.forEach((x) => visit(x, depth)); return; }
      switch (node.k) {
        case 'Func': drop(members(node.body)); return;
        case 'VarDecl': node.decls.forEach((d) => { if (d.init) assignTo(d.name, '=', d.init, depth); }); return;
        case 'ExprStmt':
          if (node.expr.k === 'Assign' && node.expr.target.k === 'Id') { assignTo(node.expr.target.name, node.expr.op, node.expr.value, depth); return; }
          drop(members(node.expr)); return;
        case 'Return': if (node.arg && linear(node.arg) === null) drop(members(node.arg)); if (node.arg && depth > 0) drop(members(node.arg)); return;
        case 'If': drop(members(node.test)); visit(node.cons, depth); visit(node.alt, depth); return;
        case 'For': visit(node.init, depth); if (node.test) drop(members(node.test)); if (node.update) drop(members(node.update)); visit(node.body, depth + 1); return;
        case 'Block': visit(node.body, depth); return;
        default: drop(members(node));
      }
    };
    const assignTo = (name, op, value, depth) => {
      if (!C.has(name)) { drop(members(value)); return; }
      const lin = linear(value);
      if (lin === null) { drop(members(value)); return; }
      if (depth > 0) { if (op === '=') drop([name]); if (lin) drop(members(value)); }        // inside loops: only += / -= of accumulator-free terms
    };
    visit(body.body, 0);
    // (3) a member accumulates, or collects other members
    for (const nm of Array.from(C)) {
      const w = writes[nm];
      if (!w.some((x) => x.op !== '=' || members(x.value).length)) drop([nm]);
    }
  }
  if (!C.size) return C;
  // does a return depend on a member?  (otherwise there is nothing to share between the lanes)
  let used = false;
  walk(body, (x) => { if (x.k === 'Return' && x.arg && members(x.arg).length) used = true; });
  return used ? C : new Set();
};

// code of an expression that is linear in the accumulators: accumulator-free addends count once (lane 0), factors stay whole
Translator.prototype.linearCode = function (e) {
  const has = (n) => { for (const nm of idsOf(n)) if (this.accSet.has(nm)) return true; return false; };
  if (!has(e)) return { code: this.asD(this.expr(e)), hasA: false };
  const guard = (x) => (x.hasA ? x.code : '((sub == 0) ? ' + x.code + ' : 0.0)');
  if (e.k === 'Id') return { code: this.asD(this.expr(e)), hasA: true };
  if (e.k === 'Unary' && e.op === '+') return this.linearCode(e.arg);
  if (e.k === 'Unary' && e.op === '-') return { code: '(-(' + this.linearCode(e.arg).code + '))', hasA: true };
  if (e.k === 'Binary' && (e.op === '+' || e.op === '-')) { const l = this.linearCode(e.l), r = this.linearCode(e.r); return { code: '(' + guard(l) + ' ' + e.op + ' ' + guard(r) + ')', hasA: true }; }
  if (e.k === 'Binary' && e.op === '*') {
    if (has(e.l)) { const l = this.linearCode(e.l); return { code: '(' + l.code + ' * ' + this.asD(this.expr(e.r)) + ')', hasA: true }; }
    const l = this.asD(this.expr(e.l)); return { code: '(' + l + ' * ' + this.linearCode(e.r).code + ')', hasA: true };
  }
  if (e.k === 'Binary' && e.op === '/') { const l = this.linearCode(e.l); return { code: '(' + l.code + ' / ' + this.asD(this.expr(e.r)) + ')', hasA: true }; }
  this.fail('internal: an accumulator in a non-linear position');
};

// Generates the body (declarations + statements) with a fix-point over the inferred local types.
Translator.prototype.functionBody = function (numericParams, allowSplit) {
  const body = this.ast.body;
  // ---- accumulator discipline (is the result a sum that lanes can share?)
  this.acc = null;
  this.split = false;
  const stmts = body.body;
  const last = stmts[stmts.length - 1];
  if (allowSplit && last && last.k === 'Return' && last.arg && last.arg.k === 'Id') {
    const name = last.arg.name;
    let ok = true, reads = 0;
    walk(body, (x) => {
      if (x.k === 'Assign' && x.target.k === 'Id' && x.target.name === name) { if (x.op !== '+=' && x.op !== '=') ok = false; if (idsOf(x.value).has(name)) ok = false; }
      else if (x.k === 'Update' && x.target.k === 'Id' && x.target.name === name) ok = false;
      if (x.k === 'Id' && x.name === name) reads++;
    });
    // occurrences: one per declaration-free assignment target, plus the final return
    let writes = 0, decls = 0;
    walk(body, (x) => {
      if (x.k === 'Assign' && x.target.k === 'Id' && x.target.name === name) writes++;
      if (x.k === 'VarDecl') x.decls.forEach((d) => { if (d.name === name) decls++; });
    });
    let returns = 0;
    walk(body, (x) => { if (x.k === 'Return' && x.arg && idsOf(x.arg).has(name)) returns++; });
    // `acc = expr` (also as the declaration's initialiser) is fine anywhere outside lane-split loops: lane 0 takes expr, the
    // other lanes restart from 0, which is what overwriting the running total means for the sum over lanes
    if (ok && decls <= 1 && reads === writes + returns && returns === 1) { this.acc = name; this.split = true; }
  }
  // several accumulators that are only ever combined linearly (`return log_prior + log_lik - 1e-3 * penalty`, helpers inlined into
  // their own running sums, reduce()): every lane keeps partial values of all of them; terms free of accumulators are added by lane 0
  this.accSet = new Set(this.acc ? [this.acc] : []);
  this.linear = false;
  if (allowSplit && !this.acc && !this.opts.single_accumulator) {
    const A = this.linearAccumulators(body);
    if (A.size) { this.accSet = A; this.linear = true; this.split = true; }
  }
  // names referenced at the top level of the function (outside every loop): lane-split loops may not leak into them
  this.topLevelRefs = new Set();
  const scanTop = (list) => {
    for (const st of list) {
      if (st.k === 'For') { this.topLoops.push(st); continue; }          // everything inside a loop statement counts as "inside the loop"
      if (st.k === 'Block') { scanTop(st.body); continue; }
      if (st.k === 'If') { idsOf(st.test).forEach((n) => this.topLevelRefs.add(n)); scanTop([st.cons]); if (st.alt) scanTop([st.alt]); continue; }
      if (st.k === 'VarDecl') { st.decls.forEach((d) => { if (d.init) idsOf(d.init).forEach((n) => this.topLevelRefs.add(n)); }); continue; }
      idsOf(st).forEach((n) => this.topLevelRefs.add(n));
    }
  };
  this.topLoops = [];
  scanTop(stmts);
  for (const a of this.accSet) this.topLevelRefs.delete(a);

  this.assignCount = {};
  walk(body, (x) => {
    if (x.k === 'VarDecl') x.decls.forEach((d) => { if (d.init) this.assignCount[d.name] = (this.assignCount[d.name] || 0) + 1; });
    if ((x.k === 'Assign' || x.k === 'Update') && x.target.k === 'Id') this.assignCount[x.target.name] = (this.assignCount[x.target.name] || 0) + 1;
  });
  this.loopCounters = new Set();
  walk(body, (x) => { if (x.k === 'For') { const c = this.canonicalLoop(x); if (c) this.loopCounters.add(c.name); } });
  this.forcedDouble = new Set();
  let lines;
  for (let round = 0; round < 8; round++) {
    const keepTypes = this.localTypes;
    this.localTypes = {};
    for (const p of numericParams) this.localTypes[p] = 'double';
    for (const nm of Object.keys(keepTypes)) if (keepTypes[nm] === 'double') this.forcedDouble.add(nm);
    this.aliases = {};
    this.localArrays = {};
    this.loopLabels = [];
    this.labelSeq = 0;
    this.condDepth = 0;
    this.declaredOnly = new Set();
    this.derivedFinal = this.derived.slice();
    this.derived = [];
    this.loops = [];
    this.pending = [];
    this.retype = false;
    this.tmp = 0;
    this.nSplit = 0;
    this.heavyLoop = false;
    this.oneLaneWork = 0;
    this.uniformNormLoops = 0;
    this.otherSplitLoops = 0;
    this.loopInfo = [];      // every canonical loop of this round: {name, single, start, bound, le} (the row plan's proof reads it)
    lines = [];
    for (const st of stmts) this.stmt(st, lines, '    ', { inLoop: false, split: false });
    if (!last || last.k !== 'Return') this.fail('log_post must end with a return statement');
    const stable = !this.retype && this.derivedFinal.join() === this.derived.join();
    if (stable) break;
    if (round === 7) this.fail('could not infer variable types');
  }
  const decl = [];
  for (const nm of Object.keys(this.localTypes)) {
    if (numericParams.indexOf(nm) >= 0) continue;
    decl.push('    ' + (this.localTypes[nm] === 'int' ? 'int' : 'double') + ' v_' + nm + ' = 0;');
  }
  for (const nm of this.derived) decl.push('    double dq_' + nm + ' = 0;');
  for (const nm of Object.keys(this.localArrays)) decl.push('    double ' + nm + '[' + this.localArrays[nm] + '] = {0};');
  return decl.concat(renderNorm(lines, 'inv'));
};

Translator.prototype.run = function () {
  this.inlineObjectCalls();
  let body = this.functionBody([], true);
  // drop data arrays the generated code never reads (constants folded away), renumber the rest
  {
    const used = new Set();
    for (const ln of body.concat(this.helperSources)) { const re = /\bA(\d+)\b/g; let m; while ((m = re.exec(ln))) used.add(Number(m[1])); }
    const remap = new Map();
    const kept = [];
    this.arrays.forEach((a, j) => { if (used.has(j)) { remap.set(j, kept.length); kept.push(a); } });
    if (kept.length !== this.arrays.length) {
      const ren = (ln) => ln.replace(/\bA(\d+)\b/g, (all, j) => 'A' + remap.get(Number(j)));
      body = body.map(ren);
      this.helperSources = this.helperSources.map(ren);
      this.arrays = kept;
    }
  }
  // did any loop actually get split?
  const parallel = this.split && this.nSplit > 0;
  // ---- LDS staging plan: whole arrays, in order of first use, while they fit the budget
  // 160 KB per CU, minus the per-chain stepper state of ~128 chains (24 B per component), minus slack
  const stateBytes = Math.min(24 * (this.P | 1) * 128, 65536);
  const budget = this.opts.lds_budget === undefined ? 163840 - 8192 - stateBytes : this.opts.lds_budget;
  const makePlan = (order) => {
    let off = 0;
    const plan = this.arrays.map(() => ({ lds: false, off: 0 }));
    for (const j of order) {
      const bytes = this.arrays[j].flat.length * this.arrays[j].esize;
      if (bytes > 0 && off + bytes <= budget) { plan[j] = { lds: true, off }; off += (bytes + 15) & ~15; }
    }
    return { plan, bytes: off };
  };
  const all = this.arrays.map((a, j) => j);
  // two plans: G > 1 lanes per chain (arrays in order of first use), and ONE lane per chain, where a fast-forwarded loop reads
  // its bit tables instead of the observations (tables first)
  // (the tables are read by the one-lane fast-forward alone -- `if constexpr (G == 1)` in the generated loops --: the G > 1 plan neither stages them
  // nor counts their bytes, which also feed the workgroup limit below)
  const isTab = (j) => this.arrays[j].key.indexOf('#aux:twoval:') === 0 || this.arrays[j].key.indexOf('#aux:kval:') === 0;
  const PG = makePlan(all.filter((j) => !isTab(j))), plan = PG.plan, off = PG.bytes;
  const P1 = this.oneLaneWork ? makePlan(all.filter(isTab).concat(all.filter((j) => !isTab(j)))) : PG;
  const D = this.derived.length;
  // Workgroup limit.  Loops with exp / log / ld.* calls need ~140 vector registers per lane: 256-thread workgroups (one wavefront per SIMD each),
  // several of them per CU -- unless the staged data leaves room for only ONE workgroup per CU (more than ~half of the 160 KB): then
  // 512 threads, so that every SIMD still holds two wavefronts (a lone one cannot hide its own fp64 latency: logit_n10k, 88 KB staged,
  // 1.52e7 -> 2.01e7 updates/s).  Plain arithmetic loops: up to 1024.
  const maxThreads = this.opts.max_threads || (this.heavyLoop ? (off > 73728 ? 512 : 256) : 1024);
  const rows = this.rowPlan(body);
  const tail = rows ? null : this.tailPlan(body);
  const ptail = (rows || tail) ? null : this.poisTailPlan(body);
  const src = [];
  src.push('// generated by bayes.js_amd/translate.js from the user\'s log_post closure');
  src.push('namespace amwg {');
  for (const h of this.helperSources) src.push(h);
  if (rows) {
    src.push('struct UserModel');
    src.push('#if defined(__HIPCC__) || defined(__HIPCC_RTC__)');
    src.push('    : UserRows<UserModel>      // row plan (csrc/amwg_rows.h): the closure ends in a likelihood loop with group means');
    src.push('#endif');
    src.push('{');
  } else
  src.push('struct UserModel {');
  src.push('  static constexpr bool kUser = true, kHasFast = false, kOneLanePass = false;');
  src.push('  static constexpr bool kHasBinary = ' + (this.hasBinary ? 'true' : 'false') + ';   // BinaryStepper branch of the step kernel');
  src.push('  static constexpr int kDerived = ' + D + ';');
  src.push('  static constexpr int kMaxThreads = ' + maxThreads + ';');
  src.push('#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC_RTC__)');
  const copies = (pl) => this.arrays.map((a, j) => pl[j].lds ? '{ ' + a.ctype + ' *dst = reinterpret_cast<' + a.ctype + ' *>(smem + ' + pl[j].off + '); const ' + a.ctype + ' *src = static_cast<const ' + a.ctype + ' *>(user_arr<' + j + '>(d)); for (int i = tid; i < ' + a.flat.length + '; i += nt) dst[i] = src[i]; }' : '').filter((x) => x);
  src.push('  __host__ __device__ static size_t lds_bytes(int, int, int lanes) { return lanes == 1 ? ' + P1.bytes + ' : ' + off + '; }');
  src.push('  __device__ static void stage(unsigned char *smem, const DataRef &d, int tid, int nt, int lanes) {');
  if (rows) src.push('    if (d.pad > 0) { stage_rows(smem, d, tid, nt); return; }      // row layout: the observations of the final loop one row per lane, the other arrays from global memory');
  if (P1 === PG) for (const c of copies(plan)) src.push('    ' + c);
  else {
    src.push('    if (lanes == 1) {');
    for (const c of copies(P1.plan)) src.push('      ' + c);
    src.push('    } else {');
    for (const c of copies(plan)) src.push('      ' + c);
    src.push('    }');
  }
  src.push('  }');
  if (rows) {
    src.push('  // ---- row plan: `' + rows.acc + '` ends in  for (i < ' + rows.n + ') ' + rows.acc + ' += ld.norm(A' + rows.y + '[i], state[' + rows.base + ' + A' + rows.labels + '[i]], ' + rows.sd + ')');
    src.push('  static constexpr int kRowN = ' + rows.n + ', kRowBase = ' + rows.base + ', kRowGroups = ' + rows.K + ', kRowY = ' + rows.y + ', kRowLabels = ' + rows.labels + ';');
    src.push('  static constexpr bool kRowDataMid = ' + (rows.mid ? 'true' : 'false') + ';');
    src.push('  // kRowSweep: a lane\'s accumulator at that loop reads the swept vector only as its OWN entry (entry k in a lane-split loop over all K entries, lane k) and');
    src.push('  // lane k < K has label k: ' + rows.sweepWhy);
    src.push('  static constexpr bool kRowSweep = ' + (rows.sweep ? 'true' : 'false') + ';');
    src.push('  __device__ __forceinline__ static double row_sd(const StateView &S, const DataRef &d) { (void)S; (void)d; return ' + rows.sd + '; }');
    src.push('  // the closure up to that loop: what this lane\'s accumulator holds when the loop begins (data arrays read from global memory)');
    src.push('  template <int G>');
    src.push('  __device__ static double head(const StateView &S, const DataRef &d, const unsigned char *smem, int sub) {');
    this.arrays.forEach((a, j) => { src.push('    const ' + a.ctype + ' *A' + j + ' = static_cast<const ' + a.ctype + ' *>(user_arr<' + j + '>(d));'); });
    src.push('    (void)smem; (void)sub; (void)d;');
    for (const ln of rows.head) src.push(ln);
    src.push('    return v_' + rows.acc + ';');
    src.push('  }');
    src.push('  // kRowCert: certified decisions in the row layout (csrc/amwg_rows.h; amwg_user_sweep_cert) -- the head only ever ADDS to `' + rows.acc + '`: ' + rows.certWhy);
    src.push('  static constexpr bool kRowCert = ' + (rows.cert ? 'true' : 'false') + ';');
    if (rows.cert) {
      src.push('  static constexpr bool kCertified = true, kCertifiedNeedsRows = true, kReferenceOrder = true;');
      src.push('  static constexpr int kCertifiedLanes = 64;');
      src.push('  // the head\'s value together with the magnitudes of what it adds up (this lane\'s share) and how many additions that is -- the bound on the two orders the head\'s');
      src.push('  // terms are summed in.  (Inlined at its three call sites: out of line every call spilled the ~170 registers the stepper keeps alive, and with the constants of');
      src.push('  // ld.norm / ld.unif / norm_inv folded by the translator the body is a handful of instructions per lane)');
      src.push('  template <int G>');
      src.push('  __device__ __forceinline__ static HeadPair head_pair(const StateView S, const DataRef &d, const unsigned char *smem, int sub) {');
      this.arrays.forEach((a, j) => { src.push('    const ' + a.ctype + ' *A' + j + ' = static_cast<const ' + a.ctype + ' *>(user_arr<' + j + '>(d));'); });
      src.push('    (void)smem; (void)sub; (void)d;');
      src.push('    double mag_ = 0.0, cnt_ = 0.0;');
      for (const ln of rows.headMag) src.push(ln);
      src.push('    return HeadPair{v_' + rows.acc + ', mag_, cnt_};');
      src.push('  }');
    }
  }
  if (tail) {
    // CERTIFIED TAIL (csrc/amwg_user.h norm_tail_approx; amwg_kernel.h "certified decisions"): with one lane per chain the stepper decides accept tests from
    // head + n c - S2 / den and its bound, and evaluates the closure itself (eval below: the reference's own order) where that does not decide
    const tx = P1.plan[tail.x];
    src.push('  // ---- certified tail: `' + tail.acc + '` ends in  for (i < ' + tail.n + ') ' + tail.acc + ' += ld.norm(A' + tail.x + '[i], ' + tail.mean + ', ' + tail.sd + ')');
    src.push('  static constexpr bool kCertifiedTail = true, kCertified = true;');
    src.push('  static constexpr int kCertifiedLanes = 1, kTailN = ' + tail.n + ';');
    src.push('  typedef TailApprox Approx;');
    src.push('  __device__ __forceinline__ static double tail_mean(const StateView &S, const DataRef &d) { (void)S; (void)d; return ' + tail.mean + '; }');
    src.push('  __device__ __forceinline__ static double tail_sd(const StateView &S, const DataRef &d) { (void)S; (void)d; return ' + tail.sd + '; }');
    src.push('  __device__ __forceinline__ static const double *tail_x_global(const DataRef &d) { return static_cast<const double *>(user_arr<' + tail.x + '>(d)); }');
    src.push('  __device__ __forceinline__ static const double *tail_x(const DataRef &d, const unsigned char *smem) { (void)smem; return ' +
             (tx.lds ? 'reinterpret_cast<const double *>(smem + ' + tx.off + ')' : 'tail_x_global(d)') + '; }      // (the one-lane plan of the arrays)');
    src.push('  // the closure up to that loop: what its accumulator holds when the loop begins');
    src.push('  template <int G>');
    src.push('  __device__ static double tail_head(const StateView &S, const DataRef &d, const unsigned char *smem, int sub) {');
    this.arrays.forEach((a, j) => {
      const pl = P1.plan[j];
      src.push('    const ' + a.ctype + ' *A' + j + ' = ' + (pl.lds ? 'reinterpret_cast<const ' + a.ctype + ' *>(smem + ' + pl.off + ')' : 'static_cast<const ' + a.ctype + ' *>(user_arr<' + j + '>(d))') + ';');
    });
    src.push('    (void)smem; (void)sub; (void)d;');
    for (const ln of tail.head) src.push(ln);
    src.push('    return v_' + tail.acc + ';');
    src.push('  }');
    src.push('  template <int G, int BT, class C>');
    src.push('  __device__ __forceinline__ static Approx log_post_approx(C &, const StateView &S, const ModelConsts &, const DataRef &d, const unsigned char *smem, int sub) {');
    src.push('    return norm_tail_approx<UserModel, G, BT>(S, d, smem, sub);');
    src.push('  }');
  }
  if (ptail) {
    // CERTIFIED POISSON TAIL (csrc/amwg_ptail.h; amwg_kernel.h "certified decisions"): with 16 lanes per chain the stepper decides accept tests from
    // head + sum eta y - sum e^eta - sum lfactorial(y) and its bound, and evaluates the closure in the reference's order where that does not decide
    const arrDecl = (j) => { const a = this.arrays[j], pl = plan[j];
      return '    const ' + a.ctype + ' *A' + j + ' = ' + (pl.lds ? 'reinterpret_cast<const ' + a.ctype + ' *>(smem + ' + pl.off + ')' : 'static_cast<const ' + a.ctype + ' *>(user_arr<' + j + '>(d))') + ';'; };
    src.push('  // ---- certified Poisson tail: `' + ptail.acc + '` ends in  for (i < ' + ptail.n + ') { ...; ' + ptail.acc + ' += ld.pois(A' + ptail.y + '[i], Math.exp(eta)) }');
    src.push('  static constexpr bool kPoisTail = true, kCertified = true, kReferenceOrder = true;');
    src.push('  static constexpr int kCertifiedLanes = 16, kTailN = ' + ptail.n + ', kStateN = ' + this.P + ';');
    src.push('  static constexpr bool kTailUniformState = ' + (ptail.uniform && this.P <= 12 ? 'true' : 'false') + ';      // the entries of the state the loop reads do not depend on the observation (and are few): scalar registers');
    src.push('  typedef TailApprox Approx;');
    src.push('  __device__ __forceinline__ static double ptail_sum_y() { return ' + hexFloat(ptail.sumY) + '; }      // sum y[i] = ' + ptail.sumY);
    src.push('  __device__ __forceinline__ static double ptail_sum_lf() { return ' + hexFloat(ptail.sumLF) + '; }      // sum lfactorial(y[i]) = ' + ptail.sumLF);
    src.push('  __device__ __forceinline__ static double ptail_y(const DataRef &d, const unsigned char *smem, int i) { (void)d; (void)smem;');
    src.push(arrDecl(ptail.y));
    src.push('    return (double)A' + ptail.y + '[i]; }');
    src.push('  __device__ __forceinline__ static double ptail_lf(const DataRef &d, const unsigned char *smem, int i) { (void)d; (void)smem;');
    src.push(arrDecl(ptail.lf));
    src.push('    return A' + ptail.lf + '[i]; }');
    src.push('  // the loop\'s statements up to the term: eta of observation v_' + ptail.i + ' for the chain whose state is S');
    src.push('  template <class SV>');
    src.push('  __device__ __forceinline__ static double ptail_eta(const SV &S, const DataRef &d, const unsigned char *smem, const int v_' + ptail.i + ') {');
    this.arrays.forEach((a, j) => src.push(arrDecl(j)));
    src.push('    (void)smem; (void)d; (void)S;');
    for (const dl of ptail.decls) src.push('    ' + dl);
    src.push('    ' + ptail.body);
    src.push('    return ' + ptail.eta + ';');
    src.push('  }');
    src.push('  static constexpr bool kTailRows = ' + (ptail.rows ? 'true' : 'false') + ';      // every data read of the statements addresses the observation\'s own row: loaded a round ahead (ptail_load)');
    if (ptail.rows) {
      const per = ptail.rows.per, js = Object.keys(per).map(Number).sort((a, b) => a - b);
      src.push('  struct TailRow { ' + js.map((j) => this.arrays[j].ctype + ' a' + j + '[' + (per[j].hi - per[j].lo) + '];').join(' ') + ' };');
      src.push('  __device__ __forceinline__ static void ptail_load(const DataRef &d, const unsigned char *smem, const int v_' + ptail.i + ', TailRow &R) {');
      for (const j of js) src.push(arrDecl(j));
      src.push('    (void)smem; (void)d;');
      for (const j of js) {
        src.push('#pragma unroll');
        src.push('    for (int q_ = 0; q_ < ' + (per[j].hi - per[j].lo) + '; ++q_) R.a' + j + '[q_] = A' + j + '[v_' + ptail.i + ' * ' + per[j].stride + ' + ' + per[j].lo + ' + q_];');
      }
      src.push('  }');
      src.push('  __device__ __forceinline__ static double ptail_y_row(const TailRow &R) { return (double)R.a' + ptail.y + '[0]; }');
      src.push('  template <class SV>');
      src.push('  __device__ __forceinline__ static double ptail_eta_row(const SV &S, const TailRow &R, const int v_' + ptail.i + ') {');
      src.push('    (void)S; (void)R;');
      for (const dl of ptail.decls) src.push('    ' + dl);
      src.push('    ' + ptail.rows.body);
      src.push('    return ' + ptail.rows.eta + ';');
      src.push('  }');
    }
    src.push('  static constexpr bool kTailLinear = ' + (ptail.linear ? 'true' : 'false') + ';      // eta is a sum of (row entry) x (state entry) products, state entries and literals: formed by fused steps');
    if (ptail.linear) {
      src.push('  static constexpr int kTailLinearRoundings = ' + ptail.linear.roundings + ';      // roundings of eta in the closure\'s statements + in the fused form');
      src.push('  template <class SV>');
      src.push('  __device__ __forceinline__ static double ptail_eta_fused(const SV &S, const TailRow &R, const int v_' + ptail.i + ') {');
      src.push('    (void)S; (void)R;');
      for (const dl of ptail.decls) src.push('    ' + dl);
      src.push('    ' + ptail.linear.body);
      src.push('    return ' + ptail.rows.eta + ';');
      src.push('  }');
      src.push('  // H >= sum of the magnitudes of eta\'s summands for every observation (column maxima of the data x |state entry|): bounds |eta| and both etas\' roundings');
      src.push('  __device__ __forceinline__ static double ptail_hlin(const StateView &S) { return ' + ptail.linear.hlin + '; }');
    }
    src.push('  // the closure up to that loop: what its accumulator holds when the loop begins (this lane\'s share)');
    src.push('  template <int G>');
    src.push('  __device__ static double tail_head(const StateView &S, const DataRef &d, const unsigned char *smem, int sub) {');
    this.arrays.forEach((a, j) => src.push(arrDecl(j)));
    src.push('    (void)smem; (void)sub; (void)d;');
    for (const ln of ptail.head) src.push(ln);
    src.push('    return v_' + ptail.acc + ';');
    src.push('  }');
    src.push('  __device__ static double ptail_head_sequence(const StateView &S, const DataRef &d, const unsigned char *smem) { return tail_head<1>(S, d, smem, 0); }      // (one lane\'s walk: the reference\'s order)');
    src.push('  // ... with the magnitudes of what it adds up and the number of additions');
    src.push('  template <int G>');
    src.push('  __device__ __forceinline__ static HeadPair ptail_head(const StateView &S, const DataRef &d, const unsigned char *smem, int sub) {');
    this.arrays.forEach((a, j) => src.push(arrDecl(j)));
    src.push('    (void)smem; (void)sub; (void)d;');
    src.push('    double mag_ = 0.0, cnt_ = 0.0;');
    for (const ln of ptail.headMag) src.push(ln);
    src.push('    return HeadPair{v_' + ptail.acc + ', mag_, cnt_};');
    src.push('  }');
    src.push('  template <int G, int BT, class C>');
    src.push('  __device__ __forceinline__ static Approx log_post_approx(C &, const StateView &S, const ModelConsts &, const DataRef &d, const unsigned char *smem, int sub) {');
    src.push('    return pois_tail_approx<UserModel, G, BT>(S, d, smem, sub);');
    src.push('  }');
    src.push('  template <int G, class C>');
    src.push('  __device__ __forceinline__ static double reference_order(C &, const StateView &S, const ModelConsts &, const DataRef &d, const unsigned char *smem, int sub) {');
    src.push('    return pois_tail_reference<UserModel, G>(S.base, &d, smem, sub);');
    src.push('  }');
  }
  src.push('#endif');
  src.push('  template <int G, bool DERIVE>');
  src.push('  AMWG_HD static double eval(const StateView &S, const DataRef &d, const unsigned char *smem, int sub, double *dv) {');
  src.push('#if defined(__HIP_DEVICE_COMPILE__)');
  this.arrays.forEach((a, j) => {
    const where = (pl) => pl[j].lds ? 'reinterpret_cast<const ' + a.ctype + ' *>(smem + ' + pl[j].off + ')' : 'static_cast<const ' + a.ctype + ' *>(user_arr<' + j + '>(d))';
    const one = where(P1.plan), many = where(plan);
    src.push('    const ' + a.ctype + ' *A' + j + ' = ' + (one === many ? many : '(G == 1) ? ' + one + ' : ' + many) + ';');
  });
  src.push('#else');
  this.arrays.forEach((a, j) => { src.push('    const ' + a.ctype + ' *A' + j + ' = static_cast<const ' + a.ctype + ' *>(user_arr<' + j + '>(d));'); });
  src.push('#endif');
  src.push('    (void)smem; (void)sub; (void)d;');
  for (const ln of body) src.push(ln);
  src.push('  }');
  src.push('};');
  src.push('}  // namespace amwg');
  return {
    source: (this.opts.no_fold_norm_inv ? src.join('\n') : foldConstantNormInv(src.join('\n'))) + '\n',
    arrays: this.arrays.map((a) => a.flat),
    array_types: this.arrays.map((a) => a.type),
    array_keys: this.arrays.map((a) => a.key),
    derived: this.derived.slice(),
    lds_bytes: off,
    lds_bytes_one_lane: P1.bytes,
    parallel: parallel ? 1 : 0,
    max_threads: maxThreads,
    work_per_eval: this.workEstimate(),
    // instruction estimate with ONE lane per chain when that enables the exact fast-forward of a two-valued sum (0 = no such loop)
    // ... or when every lane-split loop is the wave-uniform normal pass (scalar loads: one lane per chain costs nothing extra, and it is
    // the reference's own summation order, which the host library prefers when it is priced within 12 % of the cheapest geometry)
    work_one_lane: (this.oneLaneWork || (this.uniformNormLoops > 0 && !this.otherSplitLoops)) ? this.workEstimate(true) : 0,
    P: this.P,
    // row plan (csrc/amwg_rows.h; include/amwg.h amwg_user_model::rows_*): 0 / 0 / 0 = none
    rows_n_obs: rows ? rows.n : 0,
    rows_groups: rows ? rows.K : 0,
    rows_sweep: rows && rows.sweep ? 1 : 0,
    // certified tail (csrc/amwg_user.h norm_tail_approx): observations of the closure's final constant-mean normal loop; 0 = none.  (The host library reads the
    // same fact off the generated source -- kCertifiedTail / kTailN -- so no field of amwg_user_model carries it.)
    cert_tail_n: tail ? tail.n : 0,
    rows_cert: rows && rows.cert ? 1 : 0,
    // certified Poisson tail (csrc/amwg_ptail.h): observations of the closure's final log-link Poisson loop; 0 = none (the host library reads kPoisTail / kTailN off the source)
    pois_tail_n: ptail ? ptail.n : 0,
  };
};

// The CERTIFIED TAIL of a closure (csrc/amwg_user.h norm_tail_approx): its LAST statement before `return acc` is the constant-mean normal loop forLoop() marked with
// //@TAIL; everything before is the head -- it must not return early, and the closure must not write derived quantities (they are formed by the expression's pass).
// -> {x (array index), n, acc, mean, sd, head: [lines]} or null
Translator.prototype.tailPlan = function (body) {
  if (this.derived.length || this.isHelper || this.opts.no_cert_tail || this.hasBinary) return null;
  let iB = -1, iE = -1;
  body.forEach((ln, i) => { const t = ln.trim(); if (t.indexOf('//@TAIL ') === 0) iB = i; else if (t === '//@TAIL_END') iE = i; });
  if (iB < 0 || iE < iB) return null;
  const m = /^\/\/@TAIL x=A(\d+) n=(\d+) acc=(\w+) mean=(.+) @sd=(.+)$/.exec(body[iB].trim());
  if (!m) return null;
  const tailLines = body.slice(iE + 1).map((ln) => ln.trim()).filter((t) => t && t.indexOf('//') !== 0);
  if (!(tailLines.length === 2 && /^if constexpr \(DERIVE\) \{ \(void\)dv; \}$/.test(tailLines[0]) && tailLines[1] === 'return v_' + m[3] + ';')) return null;
  const head = body.slice(0, iB);
  if (head.some((ln) => /\breturn\b|\bdv\[|\bdq_/.test(ln))) return null;
  return { x: Number(m[1]), n: Number(m[2]), acc: m[3], mean: m[4], sd: m[5], head };
};

// The CERTIFIED POISSON TAIL of a closure (csrc/amwg_ptail.h): its LAST statement before `return acc` is the log-link Poisson loop forLoop() marked with //@PTAIL;
// everything before is the head -- it must not return early, must be nothing but accumulations (headMagnitude), and the closure must not write derived quantities.
// -> {y, lf (array indices), n, acc, i, uniform, eta, body, decls, head, headMag, sumY, sumLF} or null
Translator.prototype.poisTailPlan = function (body) {
  if (!this.ptailInfo || this.derived.length || this.isHelper || this.opts.no_cert_tail || this.opts.no_pois_tail || this.hasBinary) return null;
  let iB = -1, iE = -1, nB = 0;
  body.forEach((ln, i) => { const t = ln.trim(); if (t === '//@PTAIL') { iB = i; nB++; } else if (t === '//@PTAIL_END') iE = i; });
  if (nB !== 1 || iE < iB) return null;      // (ptailInfo describes the last candidate: it must be the only one)
  const info = this.ptailInfo;
  const tailLines = body.slice(iE + 1).map((ln) => ln.trim()).filter((t) => t && t.indexOf('//') !== 0);
  if (!(tailLines.length === 2 && /^if constexpr \(DERIVE\) \{ \(void\)dv; \}$/.test(tailLines[0]) && tailLines[1] === 'return v_' + info.acc + ';')) return null;
  const head = body.slice(0, iB);
  if (head.some((ln) => /\breturn\b|\bdv\[|\bdq_/.test(ln))) return null;
  const hm = this.headMagnitude(head, info.acc);
  if (hm.why) return null;
  // the closure's locals (declared at the top of the body): the loop's statements use them as scratch
  const decls = [];
  for (const ln of head) { const m = /^(double|int) (v_\w+) = 0;$/.exec(ln.trim()); if (m && m[2] !== 'v_' + info.i) decls.push(ln.trim()); }
  // sum y (integers: exact) and sum lfactorial(y) (Neumaier's compensated sum: within an ulp or two of the real sum of the stored values)
  let sumY = 0, sF = 0, cF = 0;
  const yf = this.arrays[info.y].flat, lf = this.arrays[info.lf].flat;
  for (let i = 0; i < info.n; i++) {
    sumY += yf[i];
    const t = sF + lf[i];
    cF += Math.abs(sF) >= Math.abs(lf[i]) ? (sF - t) + lf[i] : (lf[i] - t) + sF;
    sF = t;
  }
  return Object.assign({}, info, { decls, head, headMag: hm.mag, sumY, sumLF: sF + cF });
};

// The head of a plan (everything before the closure's final loop) as value + MAGNITUDES: the certified values bound the two orders the head's terms are summed in
// through the magnitudes of what it adds up, which needs the head to be nothing but accumulations -- every line that mentions the accumulator must be its
// declaration, `acc = (sub == 0) ? X : 0.0;`, `[if (sub == 0)] acc += X;` (also inside the lane-split loops), the save / restore around a loop's slow replay.
// -> {why ('' = fine), mag: the same text with every added value's magnitude summed into mag_ and the additions counted in cnt_}
Translator.prototype.headMagnitude = function (head, acc) {
  const accV = 'v_' + acc;
  const esc = accV.replace(/[$]/g, '\\$');
  const okLine = new RegExp('^(?:double ' + esc + ' = 0;|' + esc + ' = \\(sub == 0\\) \\? .+ : 0\\.0;|(?:if \\(sub == 0\\) )?' + esc + ' \\+= .+;|const double acc_save_ = ' + esc + ';|' + esc + ' = acc_save_;|for \\(.*\\) ' + esc + ' \\+= [^;]+;|for \\(.*\\) \\{ .*' + esc + ' \\+= [^;]+; \\})$');
  let why = '';
  const mag = [];
  for (const ln0 of head) {
    const t = ln0.trim();
    if (why) break;
    if (t.indexOf(accV) < 0 || t.indexOf('//') === 0) { mag.push(ln0); continue; }
    if (!okLine.test(t)) { why = 'no: the head does more with ' + acc + ' than add to it: `' + t.slice(0, 70) + '`'; break; }
    // (value, magnitude and count in ONE walk over the head's statements)
    let m2 = ln0;
    m2 = m2.replace(new RegExp(esc + ' = \\(sub == 0\\) \\? (.+) : 0\\.0;'), (q, X) => '{ const double x_ = (sub == 0) ? (' + X + ') : 0.0; ' + accV + ' = x_; mag_ = __builtin_fabs(x_); if (sub == 0) cnt_ += 1.0; }');
    m2 = m2.replace(new RegExp(esc + ' \\+= ([^;]+);', 'g'), (q, X) => '{ const double x_ = ' + X + '; ' + accV + ' += x_; mag_ += __builtin_fabs(x_); cnt_ += 1.0; }');
    m2 = m2.replace(new RegExp('const double acc_save_ = ' + esc + ';'), 'const double acc_save_ = ' + accV + ', mag_save_ = mag_;');
    m2 = m2.replace(new RegExp(esc + ' = acc_save_;'), accV + ' = acc_save_; mag_ = mag_save_;');
    mag.push(m2);
  }
  return { why, mag };
};

// The ROW PLAN of a closure (csrc/amwg_rows.h): its LAST statement before `return acc` is the gathered normal loop forLoop() marked with //@ROWS, over all
// observations, with labels that repeat with a stride of 64.  Everything before is the `head` -- it must not return early or write derived quantities.
// -> {y, labels, base, K (entries of the swept vector), n, mid, acc, sd, head: [lines], sweep, sweepWhy} or null
Translator.prototype.rowPlan = function (body) {
  if (this.derived.length || this.isHelper || this.opts.no_row_plan) return null;
  let iB = -1, iE = -1;
  body.forEach((ln, i) => { const t = ln.trim(); if (t.indexOf('//@ROWS ') === 0) iB = i; else if (t === '//@ROWS_END') iE = i; });
  if (iB < 0 || iE < iB) return null;
  const m = /^\/\/@ROWS y=A(\d+) labels=A(\d+) base=(\d+) ng=(\d+) n=(\d+) mid=([01]) acc=(\w+) sd=(.+)$/.exec(body[iB].trim());
  if (!m) return null;
  const tail = body.slice(iE + 1).map((ln) => ln.trim()).filter((t) => t && t.indexOf('//') !== 0);
  if (!(tail.length === 2 && /^if constexpr \(DERIVE\) \{ \(void\)dv; \}$/.test(tail[0]) && tail[1] === 'return v_' + m[7] + ';')) return null;
  const head = body.slice(0, iB);
  if (head.some((ln) => /\breturn\b|\bdv\[|\bdq_/.test(ln))) return null;
  const base = Number(m[3]), ng = Number(m[4]);
  // the swept vector: the parameter whose entries the labels select
  let K = 0, vec = null;
  for (const nm of Object.keys(this.layout)) { const L = this.layout[nm]; if (L.base === base && L.dim.length === 1 && !L.scalar) { K = L.len; vec = nm; } }
  if (!vec || ng > K || K > 64) return null;
  const plan = { y: Number(m[1]), labels: Number(m[2]), base, K, n: Number(m[5]), mid: m[6] === '1', acc: m[7], sd: m[8], head, sweep: false, sweepWhy: '' };
  // ---- may the proposals of a whole sweep over the vector be evaluated at once?  (a) lane k < K has label k;
  const gl = this.arrays[plan.labels].flat;
  let why = '';
  for (let l = 0; l < Math.min(K, 64, plan.n) && !why; l++) if (gl[l] !== l) why = 'no: observation ' + l + ' has label ' + gl[l];
  // (b) every read of the state in the head (and in helpers) is a scalar outside the vector, an entry of ANOTHER parameter vector, or `vec[k]` with k the
  // counter of lane-split loops over exactly 0 .. K-1 (lane k then reads entry k and no other)
  // (the loop's sd is read off the state with every proposal of the sweep written into it -- UserRows::prefetch_rows --: it is scanned like the head.  A use of
  // the state that is not an `S(index)` read -- the state handed on as a whole: an EARLIER gathered loop `norm_data_loop_gather(y, labels, S, b, ng, ...)`, a
  // helper taking the state -- is analysed if it is that call and its entries [b, b + ng) lie outside the vector, and refuses the sweep otherwise: whatever the
  // scan does not recognise is a "no".  Round-5 advisor finding: such a loop over theta[g2[i]] and an sd of |theta[0]| + 1 both came back "proved".)
  const texts = head.concat(this.helperSources || []).concat(['(row_sd) ' + plan.sd]);
  const bases = Object.keys(this.layout).map((nm) => this.layout[nm]);
  const isId = (ch) => ch !== undefined && /[\w$]/.test(ch);
  for (const ln0 of texts) {
    const cut = ln0.indexOf('//');
    const ln = cut >= 0 ? ln0.slice(0, cut) : ln0;
    // every occurrence of the identifier S that is not the read S( ... )
    for (let q = 0; !why && (q = ln.indexOf('S', q)) >= 0; q++) {
      if (isId(ln[q - 1]) || ln[q - 1] === '.' || isId(ln[q + 1])) continue;      // (part of a longer identifier, or a member)
      let r = q + 1;
      while (ln[r] === ' ') r++;
      if (ln[r] === '(') continue;                                                  // a read: checked below
      let call = null;
      const re = /norm_data_loop_gather<[^;]*?>\(A\d+, A\d+, S, (\d+), (\d+), /g;
      for (let m2; (m2 = re.exec(ln));) if (m2.index < q && q < m2.index + m2[0].length && ln.slice(q - 2, q + 3) === ', S, ') call = m2;
      if (call) {
        const b = Number(call[1]), ng = Number(call[2]);
        if (b < base + K && b + ng > base) why = 'no: an earlier gathered loop of the head reads ' + vec + ' through other labels (entries ' + (b - base) + ' .. ' + (b + ng - 1 - base) + ')';
        continue;
      }
      why = 'no: the state is handed on as a whole in `' + ln.trim().slice(0, 80) + '`';
    }
    let at = 0;
    while (!why && (at = ln.indexOf('S(', at)) >= 0) {
      if (at > 0 && /[\w.]/.test(ln[at - 1])) { at += 2; continue; }      // (another identifier that ends in S)
      let depth = 1, j = at + 2;
      while (j < ln.length && depth > 0) { if (ln[j] === '(') depth++; else if (ln[j] === ')') depth--; j++; }
      const arg = ln.slice(at + 2, j - 1).trim();
      at = j;
      let mm;
      if (/^\d+$/.test(arg)) { const c = Number(arg); if (c >= base && c < base + K) why = 'no: ' + (ln0.indexOf('(row_sd) ') === 0 ? 'the sd of the final loop' : 'the head') + ' reads entry ' + (c - base) + ' of ' + vec + ' by a constant index'; continue; }
      if ((mm = /^(?:(\d+) \+ )?v_(\w+)$/.exec(arg))) {
        const b = mm[1] ? Number(mm[1]) : 0;
        const other = bases.find((L) => L.base === b && !L.scalar);
        if (b !== base) { if (!other || (b < base + K && b + other.len > base)) why = 'no: the head reads the state at ' + arg; continue; }
        const loops = (this.loopInfo || []).filter((q) => q.name === mm[2]);
        if (!loops.length || !loops.every((q) => q.single && q.start === 0 && q.bound === K && !q.le)) why = 'no: ' + vec + '[' + mm[2] + '] is read outside a lane-split loop over 0 .. ' + (K - 1);
        continue;
      }
      why = 'no: the head reads the state at S(' + arg + ')';
    }
    if (why) break;
  }
  plan.sweep = !why && !this.hasBinary;
  plan.sweepWhy = why || (this.hasBinary ? 'no: the model has binary parameters' : 'proved');
  // ---- CERTIFIED VALUES for the row plan (round 6; csrc/amwg_rows.h UserRows::log_post_approx / sweep_approx / reference_order).  The cheap value of a lane is
  // head_l + n_l c - S2_l / den; the reference's expression sums the head's terms and the observations' in ONE running sum.  The two orders of the head's terms are
  // bounded through the MAGNITUDES of what the head adds up, which needs the head to be nothing but accumulations: every line that mentions the accumulator must be its
  // declaration, `acc = (sub == 0) ? X : 0.0;`, `[if (sub == 0)] acc += X;` (also inside the lane-split loops), the save / restore around a loop's slow replay, or the
  // return.  head_mag is the same text with every added value replaced by its magnitude and a count of the additions beside it.
  const hm = this.headMagnitude(head, plan.acc);
  let certWhy = plan.sweep ? hm.why : 'no: the sweep is not proved';
  const mag = hm.mag;
  // (head_pair is instantiated for 64 lanes only.  A lane-split loop over at most 64 entries -- the proved sweep has one over the K <= 64 entries of the swept vector --
  // gives a lane at most ONE iteration: its eight-wide block loop can never run and its remainder loop runs at most once.  Dropping the former from this copy of the
  // text keeps the certified kernel's hot path small: three inlined heads per step.)
  if (!certWhy && !this.opts.no_slim_head) {
    const slim = [];
    let short = false, skipDepth = -1;
    for (const ln of mag) {
      const t = ln.trim();
      const mN = /^const int i0_ = (\d+) \+ sub, n_ = \((\d+) - i0_ \+ G - 1\) \/ G;$/.exec(t);
      if (mN) short = Number(mN[2]) - Number(mN[1]) <= 64;
      if (skipDepth >= 0) {      // inside a dropped block loop: up to the line that closes it (same indentation as its `for`)
        if (ln.length - ln.trimStart().length === skipDepth && t === '}') skipDepth = -1;
        continue;
      }
      if (short && /^for \(; it_ \+ 8 <= n_; it_ \+= 8\) \{$/.test(t)) { skipDepth = ln.length - ln.trimStart().length; continue; }
      slim.push(ln);
    }
    mag.length = 0;
    for (const ln of slim) mag.push(ln);
  }
  plan.cert = !certWhy && !this.opts.no_row_cert;
  plan.certWhy = certWhy || (this.opts.no_row_cert ? 'no: switched off (no_row_cert)' : 'yes');
  plan.headMag = mag;
  return plan;
};

// Rough instruction count of one evaluation (steers only the lanes-per-chain choice of the host library): operators 1,
// Math.exp/log 30, pow 80, ld.norm 10 (hoisted form), other ld.* 70, loops multiply by their trip count when it is a
// translation-time constant (else by 8).
Translator.prototype.workEstimate = function (oneLane) {
  const tripOf = (st) => {
    const c = this.canonicalLoop(st);
    if (!c) return 8;
    try {
      const saved = this.pending; this.pending = [];
      const a = this.expr(c.startAst), b = this.expr(c.boundAst);
      this.pending = saved;
      if (a.cst !== undefined && b.cst !== undefined) return Math.max(0, b.cst - a.cst + (c.le ? 1 : 0));
    } catch (e) { /* not a constant */ }
    return 8;
  };
  const weigh = (node) => {
    if (!node || typeof node !== 'object') return 0;
    if (Array.isArray(node)) return node.reduce((t, x) => t + weigh(x), 0);
    let w = 0;
    switch (node.k) {
      case 'For':
        if (oneLane && node.fastForwardOneLane) return 400 * (1 + Math.log2(tripOf(node) + 2));
        return weigh(node.init) + tripOf(node) * (2 + weigh(node.test) + weigh(node.update) + weigh(node.body));
      case 'Binary': case 'Unary': case 'Assign': case 'Update': case 'Index': case 'Cond': w = 1; break;
      case 'Call': {
        const c = node.callee;
        if (c.k === 'Member' && c.obj.k === 'Id' && c.obj.name === 'Math') w = c.prop === 'pow' ? 80 : (c.prop === 'exp' || c.prop === 'log' ? 30 : 4);
        else if (c.k === 'Member' && c.obj.k === 'Id' && c.obj.name === 'ld') w = c.prop === 'norm' ? 10 : (c.prop === 'bern' ? 4 : 70);
        else w = 40;
        break;
      }
      default: break;
    }
    for (const key of Object.keys(node)) if (key !== 'k') w += weigh(node[key]);
    return w;
  };
  return weigh(this.ast.body);
};

/** translate(log_post, completedParams, data[, options]) -> {source, arrays, derived, lds_bytes, parallel, max_threads} */
function translate(fn, params, data, options) {
  return new Translator(fn, params, data, options).run();
}

module.exports = { translate, parseFunctionSource, hexFloat, tokenize, foldConstantNormInv };
