'use strict';
/*
 * models.js -- maps the user's log_post(state, data) closure (mcmc.js:958-960) to a GPU model.
 *
 * A GPU cannot run an arbitrary JS closure, so the closure has to be one of the built-in
 * families of include/amwg.h.  Two ways to say which:
 *   1. models.normal(...) / models.beta_bern(...) / models.hier_normal(...) / models.pois_glm(...)
 *      return an ordinary log_post function (it evaluates on the host with ld.js) that carries
 *      a `.amwg` tag {family, hyper, extract(data)};
 *   2. a plain closure written in the README's design pattern (README.md:149-164: `lp = 0`,
 *      `lp += ld.X(state.p, literals...)` priors, one `for` loop over the data adding one ld.*
 *      term per observation, `return lp`) is recognised from its SOURCE TEXT -- the README's own
 *      Normal (README.md:26-36) and beta-Bernoulli (README.md:150-163) examples run unchanged.
 * Anything else is refused with an explanatory string (no CPU fallback).
 */
const ld = require('./ld.js');

const FAMILY_ID = { normal: 1, beta_bern: 2, hier_normal: 3, pois_glm: 4 };
const DEFAULT_HYPER = { normal: [0, 100, 0, 100], beta_bern: [2, 2], hier_normal: [0, 100, 0, 100, 10], pois_glm: [0, 10] };

function tag(fn, family, hyper, extract, paramNames) {
  Object.defineProperty(fn, 'amwg', { value: { family, hyper, extract, paramNames }, enumerable: false });
  return fn;
}
const asArray = (d, key) => (Array.isArray(d) || ArrayBuffer.isView(d)) ? d : d[key];

// ---- explicit descriptors -------------------------------------------------------------------
function normal(opt) {
  opt = opt || {};
  const h = [].concat(opt.prior_mu || [0, 100], opt.prior_sigma || [0, 100]);
  const names = opt.names || ['mu', 'sigma'], key = opt.data_key || 'x';
  return tag(function (s, d) {
    const x = asArray(d, key);
    let lp = 0;
    lp += ld.norm(s[names[0]], h[0], h[1]);
    lp += ld.unif(s[names[1]], h[2], h[3]);
    for (let i = 0; i < x.length; i++) lp += ld.norm(x[i], s[names[0]], s[names[1]]);
    return lp;
  }, 'normal', h, (d) => ({ x: asArray(d, key) }), names);
}
function beta_bern(opt) {
  opt = opt || {};
  const h = opt.prior || [2, 2], names = opt.names || ['theta'], key = opt.data_key || 'x';
  return tag(function (s, d) {
    const x = asArray(d, key);
    let lp = 0;
    lp += ld.beta(s[names[0]], h[0], h[1]);
    for (let i = 0; i < x.length; i++) lp += ld.bern(x[i], s[names[0]]);
    return lp;
  }, 'beta_bern', h, (d) => ({ x: asArray(d, key) }), names);
}
function hier_normal(opt) {   // data {y, g, G}; params {theta:{dim:[G]}, mu:{}, sigma:{lower:0}}
  opt = opt || {};
  const h = [].concat(opt.prior_mu || [0, 100], opt.prior_sigma || [0, 100], [opt.tau === undefined ? 10 : opt.tau]);
  const names = opt.names || ['theta', 'mu', 'sigma'];
  return tag(function (s, d) {
    const th = s[names[0]], mu = s[names[1]], sg = s[names[2]];
    let lp = 0;
    lp += ld.norm(mu, h[0], h[1]);
    lp += ld.unif(sg, h[2], h[3]);
    for (let k = 0; k < th.length; k++) lp += ld.norm(th[k], mu, h[4]);
    for (let i = 0; i < d.y.length; i++) lp += ld.norm(d.y[i], th[d.g[i]], sg);
    return lp;
  }, 'hier_normal', h, (d) => ({ x: d.y, g: d.g, G: d.G }), names);
}
function pois_glm(opt) {      // data {X (N*7 row-major or array of rows), y}; params {beta:{dim:[8]}, cp:{type:"int",lower:0,upper:N-1}}
  opt = opt || {};
  const h = opt.prior_beta || [0, 10], names = opt.names || ['beta', 'cp'];
  const flatX = (d) => (Array.isArray(d.X) && Array.isArray(d.X[0])) ? [].concat.apply([], d.X) : d.X;
  return tag(function (s, d) {
    const b = s[names[0]], cp = s[names[1]], X = flatX(d), N = d.y.length;
    let lp = 0;
    for (let k = 0; k < 8; k++) lp += ld.norm(b[k], h[0], h[1]);
    lp += ld.unif(cp, 0, N - 1);
    for (let i = 0; i < N; i++) {
      let eta = 0;
      for (let k = 0; k < 7; k++) eta += X[i * 7 + k] * b[k];
      if (i >= cp) eta += b[7];
      lp += ld.pois(d.y[i], Math.exp(eta));
    }
    return lp;
  }, 'pois_glm', h, (d) => ({ x: flatX(d), y: d.y, K: 7 }), names);
}

// ---- recognition of README-pattern closures from source ---------------------------------------
function tokenize(src) {
  const re = /\s+|\/\/[^\n]*|\/\*[\s\S]*?\*\/|(\d+\.?\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?)|([A-Za-z_$][\w$]*)|(\+=|\+\+|<=|>=|===|==|=>|[-+*\/<>=(){}\[\];,.])/gy;
  const out = [];
  let m;
  re.lastIndex = 0;
  while (re.lastIndex < src.length) {
    m = re.exec(src);
    if (!m) return null;                      // a character outside the restricted grammar
    if (m[1] !== undefined) out.push({ t: 'num', v: Number(m[1]) });
    else if (m[2] !== undefined) out.push({ t: 'id', v: m[2] });
    else if (m[3] !== undefined) out.push({ t: 'p', v: m[3] });
  }
  return out;
}

/** Parses `function (S, D) { ... }` in the README pattern -> {stateArg, dataArg, priors[], lik} or null. */
function parseClosure(src) {
  const tk = tokenize(src);
  if (!tk) return null;
  let i = 0;
  const peek = (v) => i < tk.length && tk[i].v === v && tk[i].t !== 'num';
  const eat = (v) => { if (peek(v)) { i++; return true; } return false; };
  const id = () => (i < tk.length && tk[i].t === 'id') ? tk[i++].v : null;
  // header: function NAME? (a, b) {   |   (a, b) => {
  if (eat('function')) { if (!peek('(')) id(); }
  if (!eat('(')) return null;
  const S = id(); let D = null;
  if (eat(',')) D = id();
  if (!S || !eat(')')) return null;
  eat('=>');
  if (!eat('{')) return null;
  const aliases = {};          // var n = data.x.length  -> aliases.n = {len: ['x']}
  let acc = null;
  const priors = []; let lik = null, ploop = null;

  function path() {            // ID(.ID)*  -> [ids]
    const first = id(); if (!first) return null;
    const p = [first];
    while (peek('.') && tk[i + 1] && tk[i + 1].t === 'id') { i++; p.push(id()); }
    return p;
  }
  function arg(loopVar) {      // number | -number | S.name | S.name[k] | D[i] | D.f[i]
    let neg = false;
    if (eat('-')) neg = true;
    if (i < tk.length && tk[i].t === 'num') { const v = tk[i++].v; return { k: 'num', v: neg ? -v : v }; }
    if (neg) return null;
    const p = path(); if (!p) return null;
    let index = null;
    if (eat('[')) {
      if (i < tk.length && tk[i].t === 'num') index = tk[i++].v;
      else {
        const q = path(); if (!q) return null;
        if (q.length === 1 && !peek('[')) index = q[0];                       // S.theta[k]
        else {                                                                // S.theta[D.g[i]]: a label looked up in the data, by the loop variable
          if (!D || q[0] !== D || q.length < 2 || !eat('[')) return null;
          const inner = id(); if (!inner || inner !== loopVar || !eat(']')) return null;
          index = { k: 'label', field: q.slice(1) };
        }
      }
      if (index === null || !eat(']')) return null;
    }
    if (p[0] === S && p.length === 2) return { k: 'state', name: p[1], index };
    if (D && p[0] === D && index === loopVar && loopVar) return { k: 'data', field: p.slice(1) };
    return null;
  }
  function call(loopVar) {     // ld.DIST(args)
    if (!(peek('ld') && tk[i + 1] && tk[i + 1].v === '.')) return null;
    i += 2; const dist = id(); if (!dist || !eat('(')) return null;
    const args = [];
    if (!peek(')')) { do { const a = arg(loopVar); if (!a) return null; args.push(a); } while (eat(',')); }
    if (!eat(')')) return null;
    return { dist, args };
  }
  function accumulate(loopVar) {   // ACC += ld.X(...) ;?
    const nm = id(); if (nm !== acc || !eat('+=')) return null;
    const c = call(loopVar); if (!c) return null;
    eat(';');
    return c;
  }
  while (i < tk.length && !peek('}')) {
    if (peek('var') || peek('let') || peek('const')) {
      i++; const nm = id(); if (!nm || !eat('=')) return null;
      if (i < tk.length && tk[i].t === 'num' && tk[i].v === 0 && acc === null) { i++; acc = nm; }
      else { const p = path(); if (!p || p[0] !== D || p[p.length - 1] !== 'length') return null; aliases[nm] = p.slice(1, -1); }
      if (!eat(';')) return null;
    } else if (eat('for')) {
      if (lik || !eat('(')) return null;
      if (!(eat('var') || eat('let'))) return null;
      const lv = id(); if (!lv || !eat('=')) return null;
      if (!(tk[i].t === 'num' && tk[i].v === 0)) return null; i++;
      if (!eat(';') || id() !== lv || !eat('<')) return null;
      // the bound: D.x.length (or an alias of it) = the loop over the observations; S.theta.length, D.G or a number = a loop over the
      // components of a parameter (the group means' prior of the hierarchical family), allowed once, before the data loop
      let bound = null, over_param = null;
      if (i < tk.length && tk[i].t === 'num') over_param = { k: 'num', v: tk[i++].v };
      else {
        bound = path(); if (!bound) return null;
        if (bound.length === 1 && aliases[bound[0]]) bound = aliases[bound[0]];
        else if (bound[0] === D && bound[bound.length - 1] === 'length') bound = bound.slice(1, -1);
        else if (bound[0] === S && bound.length === 3 && bound[2] === 'length') { over_param = { k: 'len', name: bound[1] }; bound = null; }
        else if (D && bound[0] === D && bound.length >= 2) { over_param = { k: 'field', field: bound.slice(1) }; bound = null; }
        else return null;
      }
      if (!eat(';') || id() !== lv || !eat('++') || !eat(')')) return null;
      const braced = eat('{');
      const c = accumulate(lv); if (!c) return null;
      if (braced && !eat('}')) return null;
      if (over_param) {
        if (ploop) return null;
        const a0 = c.args[0];
        if (!a0 || a0.k !== 'state' || a0.index !== lv) return null;          // the loop variable indexes the parameter: S.theta[k]
        ploop = { call: c, count: over_param };
        continue;
      }
      lik = { call: c, over: bound };
    } else if (eat('return')) {
      if (id() !== acc) return null; eat(';');
    } else {
      const c = accumulate(null); if (!c) return null;
      if (lik || ploop) return null;          // scalar priors come first, then the loop over a parameter's components, then the data loop (summation order)
      priors.push(c);
    }
  }
  if (!eat('}') || i !== tk.length || !acc || !lik) return null;
  return { priors, lik, ploop };
}

const isNum = (a) => a && a.k === 'num';
const isState = (a) => a && a.k === 'state' && a.index === null;

/** -> {family, hyper, extract, paramNames} or null */
function recognise(fn) {
  if (typeof fn !== 'function') return null;
  if (fn.amwg) return fn.amwg;
  const ast = parseClosure(Function.prototype.toString.call(fn));
  if (!ast) return null;
  const { priors, lik, ploop } = ast, L = lik.call, field = lik.over;
  const dataArg = L.args[0];
  if (!dataArg || dataArg.k !== 'data' || dataArg.field.join('.') !== field.join('.')) return null;
  const dig = (d, f) => { let v = d; for (const q of f) v = v[q]; return v; };
  const extract = (d) => ({ x: dig(d, field) });
  if (ploop) {
    // the hierarchical Normal family written out (SURVEY.md section 8(d) cfg4; the order of the terms is the kernel's: the prior of the
    // location, the prior of the scale, the group means' prior in a loop, the observations):
    //     lp += ld.norm(S.mu, m0, s0); lp += ld.unif(S.sigma, a, b);
    //     for (k < G) lp += ld.norm(S.theta[k], S.mu, tau);
    //     for (i < D.y.length) lp += ld.norm(D.y[i], S.theta[D.g[i]], S.sigma);
    const Q = ploop.call;
    if (priors.length !== 2 || L.dist !== 'norm' || L.args.length !== 3 || Q.dist !== 'norm' || Q.args.length !== 3) return null;
    const [p0, p1] = priors, mean = L.args[1], sd = L.args[2], th = Q.args[0];
    if (!(p0.dist === 'norm' && p1.dist === 'unif' && p0.args.length === 3 && p1.args.length === 3 && isState(p0.args[0]) && isState(p1.args[0]) &&
          isNum(p0.args[1]) && isNum(p0.args[2]) && isNum(p1.args[1]) && isNum(p1.args[2]))) return null;
    if (!(mean && mean.k === 'state' && mean.index && mean.index.k === 'label' && isState(sd) && sd.name === p1.args[0].name)) return null;
    if (!(th.name === mean.name && isState(Q.args[1]) && Q.args[1].name === p0.args[0].name && isNum(Q.args[2]))) return null;
    if (ploop.count.k === 'len' && ploop.count.name !== th.name) return null;
    const gField = mean.index.field, count = ploop.count;
    return { family: 'hier_normal', hyper: [p0.args[1].v, p0.args[2].v, p1.args[1].v, p1.args[2].v, Q.args[2].v],
             paramNames: [th.name, p0.args[0].name, p1.args[0].name],
             // the loop bound is the number of group means the closure adds prior terms for: it must be the parameter's dim (checked by the caller
             // through G against the declared params; a closure that loops over fewer components is not this family)
             extract: (d, params) => {
               const G = count.k === 'num' ? count.v : (count.k === 'field' ? dig(d, count.field) : (params && params[th.name] ? params[th.name].dim[0] : undefined));
               return { x: dig(d, field), g: dig(d, gField), G };
             },
             dimCheck: { name: th.name, count } };
  }
  if (L.dist === 'norm' && L.args.length === 3 && isState(L.args[1]) && isState(L.args[2]) && priors.length === 2) {
    const [p0, p1] = priors;   // the closure's own order: norm prior on the mean, then unif prior on the sd
    if (p0.dist === 'norm' && p1.dist === 'unif' && isState(p0.args[0]) && isState(p1.args[0]) &&
        p0.args[0].name === L.args[1].name && p1.args[0].name === L.args[2].name &&
        isNum(p0.args[1]) && isNum(p0.args[2]) && isNum(p1.args[1]) && isNum(p1.args[2]) && p0.args.length === 3 && p1.args.length === 3)
      return { family: 'normal', hyper: [p0.args[1].v, p0.args[2].v, p1.args[1].v, p1.args[2].v], extract,
               paramNames: [L.args[1].name, L.args[2].name] };
    return null;
  }
  if (L.dist === 'bern' && L.args.length === 2 && isState(L.args[1]) && priors.length === 1) {
    const p0 = priors[0];
    if (p0.dist === 'beta' && p0.args.length === 3 && isState(p0.args[0]) && p0.args[0].name === L.args[1].name &&
        isNum(p0.args[1]) && isNum(p0.args[2]))
      return { family: 'beta_bern', hyper: [p0.args[1].v, p0.args[2].v], extract, paramNames: [L.args[1].name] };
  }
  return null;
}

module.exports = { normal, beta_bern, hier_normal, pois_glm, recognise, parseClosure, FAMILY_ID, DEFAULT_HYPER };
