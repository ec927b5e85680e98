'use strict';
// fuzz_translate_cli.js -- test helper (no GPU): random closures for the translator.
//   node tests/js/fuzz_translate_cli.js <outdir> <seed> <n_models> [n_derived = 48]
// Every model is one closure with 48 derived quantities `s.qK = <random expression>` (arithmetic, comparisons, ?:, && ||, Math.*, ld.*,
// integer and double operands, -0, NaN, Infinity), a few random statement blocks (loops over the data with if/else, continue, local
// arrays, integer counters) and a random return expression.  The closure is evaluated by V8 at 40 random states (with this package's
// ld.js, itself pinned bit for bit against the reference's distributions.js) and translated with the PRODUCT's translator; the test
// (tests/test_translate.py) compiles the generated text for the host and compares all 49 values per state bit for bit.
// writes <outdir>/fuzz_<seed>_<k>.{hip,arrays.bin,meta.json,states.json,js}
const fs = require('fs');
const path = require('path');
const { mcmc, ld } = require('../../bayes.js_amd');
global.ld = ld;
const out = process.argv[2], seed0 = Number(process.argv[3] || 1), nModels = Number(process.argv[4] || 3), NQ_ARG = Number(process.argv[5] || 48);

function rng(seed) { let s = seed >>> 0; return () => { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; return s / 4294967296; }; }
const bits = (v) => { const b = Buffer.alloc(8); b.writeDoubleBE(v); return b.toString('hex'); };

function generator(rnd) {
  const pick = (a) => a[Math.floor(rnd() * a.length)];
  const lit = () => pick(['Number.EPSILON', 'Number.MAX_SAFE_INTEGER', '0', '1', '2', '3', '(-1)', '0.5', '(-2.5)', '1e-3', '7', '10', '0.1', '1.5', '100', '(-0)', '3.25', '1e10', '4', '6', '0.25']);
  // leaves: real params a, b (b > 0), int param k in 0..6, binary z, vector v[3]; data x[8] doubles, n[8] small ints, m[2][3] doubles
  // reads that may fall outside the array or use a non-integer index: undefined in JavaScript, NaN in arithmetic
  const wild = (ctx) => pick(['d.x[s.k + 4]', 'd.x[s.k - 2]', 'd.x[s.a]', 'd.x[s.v[0] * 2]', 's.v[d.n[' + Math.floor(rnd() * 8) + ']]', 'd.n[d.n[0]]', 's.v[s.k]', 'd.m[1][s.k]']
    .concat(ctx.i ? ['d.x[' + ctx.i + ' + 3]', 'd.x[' + ctx.i + ' - 1]', 's.v[d.n[' + ctx.i + ']]', 'd.x[' + ctx.i + ' * 2]', 'd.x[' + ctx.i + ' / 2]'] : []));
  const leafD = (ctx) => (rnd() < 0.06 && !ctx.calm) ? wild(ctx) : pick(['s.a', 's.b', 's.v[0]', 's.v[1]', 's.v[2]', 's.w[1][2]', 's.w[s.z][s.k % 3]', 's.w[0][' + Math.floor(rnd() * 3) + ']', 'd.x[' + Math.floor(rnd() * 8) + ']', 'd.m[' + Math.floor(rnd() * 2) + '][' + Math.floor(rnd() * 3) + ']', lit(), lit()]
    .concat(ctx.i ? ['s.w[' + ctx.i + ' % 2][(' + ctx.i + ' + s.k) % 3]', 'd.x[' + ctx.i + ']', 'd.x[' + ctx.i + ']', 't', 'd.x[(' + ctx.i + ' * 3 + 1) % 8]', 'd.m[' + ctx.i + ' % 2][(' + ctx.i + ' + s.k) % 3]', 's.v[' + ctx.i + ' % 3]'] : []));
  const leafI = (ctx) => pick(['s.k', 's.z', 'd.n[' + Math.floor(rnd() * 8) + ']', String(Math.floor(rnd() * 9)), 'd.x.length'].concat(ctx.i ? [ctx.i, ctx.i, 'd.n[' + ctx.i + ']', '(' + ctx.i + ' * d.n[7 - ' + ctx.i + '])', '(d.n[' + ctx.i + '] % 3)'] : []));
  function num(depth, ctx) {
    if (depth <= 0 || rnd() < 0.18) return rnd() < 0.7 ? leafD(ctx) : leafI(ctx);
    const r = rnd();
    const a = () => num(depth - 1, ctx);
    if (r < 0.30) return '(' + a() + ' ' + pick(['+', '-', '*', '+', '-', '*', '/']) + ' ' + a() + ')';
    if (r < 0.34) return '(' + a() + ' % ' + pick(['3', '2.5', '(s.k + 1)', '7', leafI(ctx) + ' + 1']) + ')';
    if (r < 0.355) return '(- ' + a() + ')';
    if (r < 0.37) return '(' + a() + ' ' + pick(['||', '&&']) + ' ' + a() + ')';      // value-selecting, on numbers
    if (r < 0.40) return pick([() => '(' + a() + ' ' + pick(['|', '&', '^', '<<', '>>', '>>>']) + ' ' + pick([leafI(ctx), '0', '3', '31', a()]) + ')', () => '(~' + a() + ')', () => '(~~' + a() + ')',
      () => 'Math.imul(' + a() + ', ' + a() + ')', () => 'Math.clz32(' + a() + ')', () => 'Math.fround(' + a() + ')', () => '((' + a() + ' * 1e9) | 0)', () => '((' + a() + ' * 1e10) >>> 0)'])();
    if (r < 0.50) return '(' + cond(depth - 1, ctx) + ' ? ' + a() + ' : ' + a() + ')';
    if (r < 0.72) {
      const f = pick(['abs', 'floor', 'ceil', 'round', 'trunc', 'sign', 'sqrt', 'exp', 'log', 'log1p', 'expm1', 'tanh', 'atan', 'log10', 'abs', 'sqrt', 'exp', 'log',
        'sin', 'cos', 'tan', 'asin', 'acos', 'sinh', 'cosh', 'asinh', 'acosh', 'atanh', 'cbrt', 'log2']);
      if (f === 'sinh' || f === 'cosh') return 'Math.' + f + '(' + a() + ' * 0.05)';
      if (f === 'exp' || f === 'expm1') return 'Math.' + f + '(' + a() + ' * 0.1)';
      return 'Math.' + f + '(' + a() + ')';
    }
    if (r < 0.76) return 'Math.' + pick(['min', 'max']) + '(' + a() + ', ' + a() + (rnd() < 0.3 ? ', ' + a() : '') + ')';
    if (r < 0.78) return rnd() < 0.5 ? 'Math.atan2(' + a() + ', ' + a() + ')' : 'Math.hypot(' + a() + ', ' + a() + (rnd() < 0.4 ? ', ' + a() : '') + ')';
    if (r < 0.83) return 'Math.pow(' + a() + ', ' + pick(['2', '0.5', '3', '-1', '1.5', leafD(ctx), 's.k']) + ')';
    if (r < 0.97) {
      const pos = () => 'Math.abs(' + a() + ') + 0.1';
      return pick([
        () => 'ld.norm(' + a() + ', ' + a() + ', ' + pos() + ')', () => 'ld.unif(' + a() + ', -3, 9)', () => 'ld.pois(' + leafI(ctx) + ', ' + pos() + ')',
        () => 'ld.gamma(' + pos() + ', ' + pos() + ', ' + pos() + ')', () => 'ld.beta(Math.abs(Math.tanh(' + a() + ')), 2, ' + pos() + ')', () => 'ld.bern(s.z, 0.3)',
        () => 'ld.binom(' + leafI(ctx) + ', 12, Math.abs(Math.tanh(' + a() + ')))', () => 'ld.cauchy(' + a() + ', ' + a() + ', ' + pos() + ')', () => 'ld.laplace(' + a() + ', 1, ' + pos() + ')',
        () => 'ld.t(' + a() + ', ' + a() + ', ' + pos() + ', ' + pos() + ')', () => 'ld.exp(' + pos() + ', ' + pos() + ')', () => 'ld.lnorm(' + pos() + ', ' + a() + ', ' + pos() + ')',
        () => 'ld.logis(' + a() + ', ' + a() + ', ' + pos() + ')', () => 'ld.weibull(' + pos() + ', ' + pos() + ', ' + pos() + ')', () => 'ld.nbinom(' + leafI(ctx) + ', ' + pos() + ', 0.4)',
      ])();
    }
    return '(' + cond(depth - 1, ctx) + ' ? 1 : 0)';
  }
  function cond(depth, ctx) {
    // no possibly-undefined reads inside comparisons: `undefined == undefined` is true in JavaScript, NaN == NaN is not (the translator
    // stands NaN in for undefined, which is exact in arithmetic and in every comparison with a number)
    ctx = Object.assign({}, ctx, { calm: true });
    const r = rnd();
    if (depth <= 0 || r < 0.55) return '(' + num(depth - 1, ctx) + ' ' + pick(['<', '>', '<=', '>=', '===', '!==', '==', '!=']) + ' ' + num(depth - 1, ctx) + ')';
    if (r < 0.75) return '(' + cond(depth - 1, ctx) + ' ' + pick(['&&', '||']) + ' ' + cond(depth - 1, ctx) + ')';
    if (r < 0.85) return '(!' + cond(depth - 1, ctx) + ')';
    if (r < 0.90) return pick(['isNaN', 'isFinite', 'Number.isInteger', 'Number.isSafeInteger', 'Number.isNaN']) + '(' + num(depth - 1, ctx) + ')';
    if (r < 0.92 && !ctx.i && !ctx.noLoops) return pick(['d.x', 'd.n', 's.v', 's.w[0]']) + '.' + pick(['some', 'every']) + '((qq, jj) => qq ' + pick(['<', '>', '>=', '!==']) + ' ' + num(1, { calm: true, noLoops: true }) + ' + jj)';
    return '(s.z === ' + pick(['0', '1']) + ')';
  }
  function block(k) {     // statement templates around random expressions
    const ctx = { i: 'i' }, r = rnd(), q = 'acc' + k;
    const e = () => num(3, ctx), c = () => '(' + cond(2, ctx) + ')';
    if (r < 0.35) return 'var ' + q + ' = ' + lit() + ';\n  for (var i = 0; i < d.x.length; i++) { var t = ' + num(2, { }) + '; if ' + c() + ' { ' + q + ' += ' + e() + '; } else { ' + q + ' -= ' + e() + ' * 0.5; } }\n  s.b' + k + ' = ' + q + ';';
    if (r < 0.55) return 'var ' + q + ' = 0;\n  for (var i = 0; i < 8; i++) { var t = d.x[i] * ' + lit() + '; if ' + c() + ' continue; ' + q + ' += ' + e() + '; if (' + q + ' > 1e6) break; }\n  s.b' + k + ' = ' + q + ';';
    if (r < 0.75) return 'var arr' + k + ' = [' + num(2, {}) + ', ' + num(2, {}) + ', ' + num(2, {}) + '];\n  var ' + q + ' = 0, cnt' + k + ' = 0;\n  for (var i = 0; i < 3; i++) { var t = arr' + k + '[i]; if ' + c() + ' { cnt' + k + '++; ' + q + ' += arr' + k + '[(i + s.k) % 3] * ' + e() + '; } }\n  s.b' + k + ' = ' + q + ' + cnt' + k + ';';
    return 'var ' + q + ' = 1, j' + k + ' = 0;\n  while (j' + k + ' < s.k + 2) { var i = j' + k + ' % 8; var t = ' + q + '; ' + q + ' = ' + q + ' * 0.5 + ' + e() + ' * 1e-3; j' + k + ' += 1; }\n  s.b' + k + ' = ' + q + ';';
  }
  function lpBlock() {
    const ctx = { i: 'i' }, r = rnd();
    const e = () => num(3, ctx), c = () => '(' + cond(2, ctx) + ')';
    if (r < 0.3) return 'for (var i = 0; i < d.x.length; i++) { lp += ' + e() + ' * 1e-2; }';
    if (r < 0.55) return 'for (var i = 0; i < d.x.length; i++) { var t = ' + num(2, { i: 'i' }).replace(/(^|[^.\w])t\b/g, '$1s.a') + '; if ' + c() + ' { lp += ' + e() + ' * 1e-2; } else { lp -= ' + e() + ' * 1e-2; } }';
    if (r < 0.75) return 'for (var i = 0; i < 8; i++) { var t = d.x[i] - ' + lit() + '; var u = t * t; if ' + c() + ' continue; lp += (u + ' + e() + ') * 1e-2; }';
    if (r < 0.9) return 'for (var i = 0; i < d.n.length; i++) { var t = 0; for (var j = 0; j <= d.n[i] % 4; j++) { t += d.x[(i + j) % 8] * ' + lit() + '; } lp += ld.norm(t, s.a, s.b + 0.5) * 1e-2; }';
    return 'lp += ' + num(3, {}) + ' * 1e-2;';
  }
  // post-ES5 spellings (rewritten by the parser into the core subset): for-of, forEach with an early return, reduce inside an
  // expression, destructuring declarations, an arrow helper
  function sugarBlock(k) {
    const r = rnd(), ctxX = { i: null };
    const e = (ctx) => num(2, ctx || {}), c = (ctx) => '(' + cond(2, ctx || {}) + ')';
    const withT = (str, v) => str.replace(/(^|[^.\w])t\b/g, '$1' + v);
    if (r < 0.05) return 'for (var i = 0; i < 8; i++) { switch (d.n[i] % 4) { case 0: lp += ' + e({ i: 'i' }) + ' * 1e-3; break; case 1: case 2: { lp -= ' + e({ i: 'i' }) + ' * 1e-3; break; } default: lp += 1e-3; } }';
    if (r < 0.10) return 'var jd' + k + ' = 0; do { lp += ' + e() + ' * 1e-3; jd' + k + '++; } while (jd' + k + ' < s.k);';
    if (r < 0.18) return 'for (const rw of d.rows) { if (rw.tag === "u") lp += rw.val * ' + e() + ' * 1e-3; else if (rw.tag !== "w") { lp -= rw.sub.q * 1e-3; } switch (rw.tag) { case "v": lp += 1e-4; break; case "zz": lp += 1; break; default: lp -= 1e-4; } }';
    if (r < 0.22) return 'for (let i = 0; i < d.rows.length; i++) { const rw = d.rows[i]; const q = [rw.val, ' + e({ i: 'i' }) + ']; lp += (q[0] - q[1]) * rw.sub.q * 1e-3; }';    // block-scoped names reused across blocks
    if (r < 0.25) return '{ const q = ' + e() + '; let rw = q * 2; lp += (q + rw) * 1e-4; }';
    if (r < 0.29) return 'var pa' + k + ' = []; for (var i = 0; i < d.x.length; i++) { var t = d.x[i] * 2; pa' + k + '.push(' + e({ i: 'i' }) + ' + t); }\n  lp += pa' + k + '[d.n[3] % 8] * 1e-3; s.r' + k + ' = pa' + k + '[2] - pa' + k + '.length;';
    if (r < 0.33) return 'for (const xv' + k + ' of d.x) { lp += (xv' + k + ' * ' + withT(e(), 'xv' + k) + ') * 1e-3; }';
    if (r < 0.3) return 'd.x.forEach(function (xe, ie) { if ' + withT(c({ i: 'ie' }), 'xe') + ' return; lp += (xe + ' + withT(e({ i: 'ie' }), 'xe') + ') * 1e-3; });';
    if (r < 0.45) return 'lp += d.x.reduce((ac, xe, ie) => ac + ' + withT(e({ i: 'ie' }), 'xe') + ' * 1e-3, ' + withT(e(), 's.a') + ') * 1e-2;';
    if (r < 0.55) return 'const { a: pa' + k + ', v: [pv' + k + ', , pw' + k + '] } = s;\n  lp += (pa' + k + ' * pv' + k + ' - pw' + k + ') * 1e-3;';
    if (r < 0.65) return 'const hf' + k + ' = (p, q) => p * ' + lit() + ' + Math.abs(q);\n  for (const nv' + k + ' of d.n) lp += hf' + k + '(nv' + k + ', ' + withT(e(), 's.b') + ') * 1e-3;';
    if (r < 0.72) return 's.r' + k + ' = s.v.reduce(function (ac, ve) { var sq = ve * ve; return ac + sq; }, 0) + d.m[1].reduce((ac, me) => Math.max(ac, me), -Infinity) + d.x.reduce((p, q) => p + q * ' + lit() + ') - s.w[1].reduce((p, q) => Math.min(p, q));';
    return pick([
      'var mz' + k + ' = d.x.map(function (xe, ie) { return ' + withT(e({ i: 'ie' }), 'xe') + '; });\n  for (var i = 0; i < mz' + k + '.length; i++) { lp += mz' + k + '[i] * 1e-3; }\n  s.r' + k + ' = mz' + k + '[' + Math.floor(rnd() * 8) + '];',
      'lp += d.n.map((ne) => ne * ' + e() + ').reduce((ac, q) => ac + q, 0) * 1e-3;',
      'var cn' + k + ' = new Array(3).fill(' + lit() + '); for (var i = 0; i < 8; i++) { cn' + k + '[d.n[i] % 3] += d.x[i]; }\n  s.r' + k + ' = cn' + k + '[0] - cn' + k + '[1] * cn' + k + '[2];',
      's.v.map((ve) => ve * ' + e() + ').forEach((q, j) => { lp += q * (j + 1) * 1e-3; });']);
  }
  return { num, cond, block, lpBlock, sugarBlock };
}

function stateFrom(rnd, t) {
  const special = [0, -0, 1, -1, 0.5, 2, 1e-8, 30, -30, 3];
  const real = () => (rnd() < 0.15 ? special[Math.floor(rnd() * special.length)] : (rnd() - 0.5) * (rnd() < 0.3 ? 40 : 4));
  return { a: t === 0 ? 0.5 : real(), b: Math.abs(real()) + (rnd() < 0.1 ? 0 : 0.05), v: [real(), real(), real()], k: Math.floor(rnd() * 7), z: rnd() < 0.5 ? 0 : 1, w: [[real(), real(), real()], [real(), real(), real()]] };
}

const names = [];
for (let mk = 0; mk < nModels; mk++) {
  const seed = seed0 * 1000 + mk, rnd = rng(seed), G = generator(rnd);
  const lines = [];
  const NQ = NQ_ARG, NB = NQ_ARG >= 48 ? 5 : 2;     // a smaller model for the device test: hiprtc compiles it in seconds
  for (let q = 0; q < NQ; q++) lines.push('  s.q' + q + ' = ' + G.num(4, {}) + ';');
  for (let b = 0; b < NB; b++) lines.push('  ' + G.block(b));
  lines.push('  var lp = ' + G.num(3, {}) + ' + ld.norm(s.a, 0, 10);');
  for (let b = 0; b < 4; b++) lines.push('  ' + G.lpBlock());
  for (let b = 0; b < 3; b++) lines.push('  ' + G.sugarBlock(b));
  // half of the models return a linear combination of several running sums (the translator's multi-accumulator lane splitting)
  if (mk % 2 === 1) {
    const ctxI = { i: 'i' };
    lines.push('  var la0 = ' + G.num(2, {}) + ' * 1e-3, la1 = 0, la2 = 1;');
    lines.push('  for (var i = 0; i < d.x.length; i++) { la0 += ' + G.num(3, ctxI) + ' * 1e-3; }');
    lines.push('  for (var i = 0; i < 8; i++) { var t = d.x[i] * 0.5; if (' + G.cond(2, ctxI) + ') { la1 -= ' + G.num(2, ctxI) + ' * 1e-3; } else { la1 += t * 1e-3; la0 -= 1e-4; } }');
    lines.push('  for (var i = 0; i < d.n.length; i++) { la2 += d.n[i] * ' + G.num(2, {}) + ' * 1e-4; }');
    lines.push('  s.la2seen = la2;');           // la2 is read outside the linear forms: it must stay an ordinary (unsplit) variable
    lines.push('  la1 += la0 * 0.25 + ' + G.num(2, {}) + ' * 1e-3;');
    lines.push('  return lp + 0.5 * la1 - la0 / 3 + la2 * 1e-3 + ' + G.num(2, {}) + ' * 1e-3;');
  } else {
    lines.push('  return lp;');
  }
  const src = 'return function (s, d) {\n' + lines.join('\n') + '\n};';
  const fn = new Function('ld', src)(ld);
  const data = { x: [], n: [], m: [[0.5, -1.25, 3], [2, 0, -0.75]], rows: [] };
  for (let i = 0; i < 8; i++) { data.x.push(i === 3 ? 0 : (rnd() - 0.4) * 6); data.n.push(Math.floor(rnd() * 11)); data.rows.push({ val: (rnd() - 0.5) * 3, tag: ['u', 'v', 'w'][Math.floor(rnd() * 3)], sub: { q: Math.floor(rnd() * 5) } }); }
  const params = mcmc.complete_params({ a: {}, b: { lower: 0 }, v: { dim: [3] }, k: { type: 'int', lower: 0, upper: 6 }, z: { type: 'binary' }, w: { dim: [2, 3] } }, mcmc.param_init_fixed);
  const name = 'fuzz_' + seed0 + '_' + mk + (NQ_ARG === 48 ? '' : '_q' + NQ_ARG);
  fs.writeFileSync(path.join(out, name + '.js'), src);
  let tr;
  try { tr = mcmc.translate(fn, params, data, {}); } catch (e) { console.error('TRANSLATE FAILED for ' + name + ': ' + e); process.exit(3); }
  fs.writeFileSync(path.join(out, name + '.hip'), tr.source);
  let bytes = 4;
  for (const a of tr.arrays) bytes += 8 + a.length * 8;
  const buf = Buffer.alloc(bytes);
  let o = 0;
  buf.writeUInt32LE(tr.arrays.length, o); o += 4;
  for (const a of tr.arrays) { buf.writeBigUInt64LE(BigInt(a.length), o); o += 8; for (let i = 0; i < a.length; i++) { buf.writeDoubleLE(a[i], o); o += 8; } }
  fs.writeFileSync(path.join(out, name + '.arrays.bin'), buf);
  fs.writeFileSync(path.join(out, name + '.meta.json'), JSON.stringify({ name, P: tr.P, derived: tr.derived, lds_bytes: tr.lds_bytes, lds_bytes_one_lane: tr.lds_bytes_one_lane, parallel: tr.parallel,
    max_threads: tr.max_threads, work_per_eval: tr.work_per_eval, work_one_lane: tr.work_one_lane, rows_n_obs: tr.rows_n_obs, rows_groups: tr.rows_groups, rows_sweep: tr.rows_sweep, cert_tail_n: tr.cert_tail_n, rows_cert: tr.rows_cert, array_keys: tr.array_keys, array_types: tr.array_types, array_len: tr.arrays.map((a) => a.length) }));
  const pts = [];
  for (let t = 0; t < 40; t++) {
    const st = stateFrom(rnd, t);
    const flat = [st.a, st.b, st.v[0], st.v[1], st.v[2], st.k, st.z, st.w[0][0], st.w[0][1], st.w[0][2], st.w[1][0], st.w[1][1], st.w[1][2]];
    const lp = fn(st, data);
    pts.push({ state: flat.map(bits), lp: bits(lp), derived: tr.derived.map((k) => bits(st[k])) });
  }
  fs.writeFileSync(path.join(out, name + '.states.json'), JSON.stringify(pts));
  names.push(name);
}
console.log(names.join(' '));
