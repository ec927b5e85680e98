#!/usr/bin/env node
'use strict';
/*
 * bench/js_e2e.js -- the path a bayes.js user actually runs, end to end, through the JavaScript host (SURVEY.md section 8(d): kernel-only AND
 * end-to-end): `require('bayes.js_amd')`, `new mcmc.AmwgSampler(params, log_post, data, {chains})` with the README's closure (translated
 * and compiled with hiprtc, or loaded from the on-disk code-object cache), `burn(n)`, `sample(n)` INCLUDING the copy of every recorded draw
 * to host memory and into the typed arrays sample() returns (mcmc.js:1005-1030), then `close()`.
 *
 *   node bench/js_e2e.js [--chains 65536] [--n-obs 10000] [--burn 1000] [--sample 1000] [--thin 1] [--single 1]
 *
 * Prints one JSON line.  `--single 1` also times the reference's own use: ONE chain, README data, constructed twice (the second
 * construction finds the code object in the cache).  Data: a local deterministic generator (nothing under oracle/ is used here).
 */
const path = require('path');
const args = process.argv.slice(2);
const opt = (name, dflt) => { const i = args.indexOf('--' + name); return i >= 0 ? Number(args[i + 1]) : dflt; };
const chains = opt('chains', 65536), N = opt('n-obs', 10000), nBurn = opt('burn', 1000), nSample = opt('sample', 1000), thin = opt('thin', 1);
const now = () => Number(process.hrtime.bigint()) * 1e-6;     // ms

const t_req0 = now();
const { mcmc, ld } = require(path.join(__dirname, '..', 'bayes.js_amd'));
global.ld = ld;
const t_req = now() - t_req0;

// x_i ~ Normal(3, 2): xorshift32 + Box-Muller, fixed seed
function makeData(n) {
  let s = 20260925 >>> 0;
  const u = () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return (s + 0.5) / 4294967296; };
  const x = new Array(n);
  for (let i = 0; i < n; i += 2) {
    const r = Math.sqrt(-2 * Math.log(u())), a = 2 * Math.PI * u();
    x[i] = 3 + 2 * r * Math.cos(a);
    if (i + 1 < n) x[i + 1] = 3 + 2 * r * Math.sin(a);
  }
  return x;
}
const params = { mu: { type: 'real' }, sigma: { type: 'real', lower: 0 } };
const log_post = function (state, data) {          // README.md:26-36
  var log_post = 0;
  log_post += ld.norm(state.mu, 0, 100);
  log_post += ld.unif(state.sigma, 0, 100);
  for (var i = 0; i < data.length; i++) {
    log_post += ld.norm(data[i], state.mu, state.sigma);
  }
  return log_post;
};

// the same model with its priors written the other way round: NOT recognised as the family, hence translated (translate.js) and compiled (hiprtc) -- since round 6
// such a closure gets certified decisions from the translator when it ends in the constant-mean normal loop (tailPlan: amwg_user_step_cert)
const params_swapped = { sigma: { type: 'real', lower: 0 }, mu: { type: 'real' } };
const log_post_swapped = function (state, data) {
  var lp = 0;
  lp += ld.unif(state.sigma, 0, 100);
  lp += ld.norm(state.mu, 0, 100);
  for (var i = 0; i < data.length; i++) lp += ld.norm(data[i], state.mu, state.sigma);
  return lp;
};

function run(data, chainCount, burn, sample, thinBy, translated) {
  const t0 = now();
  const s = translated ? new mcmc.AmwgSampler(params_swapped, log_post_swapped, data, { chains: chainCount, seed: 20260925 })
                       : new mcmc.AmwgSampler(params, log_post, data, { chains: chainCount, seed: 20260925 });
  const t1 = now();
  s.burn(burn);
  const t2 = now();
  if (thinBy > 1) s.thin(thinBy);
  const smp = s.sample(sample);
  const t3 = now();
  const kept = Math.ceil(sample / thinBy);
  let bytes = 0, meanMu = 0;
  for (const k of Object.keys(smp)) bytes += chainCount === 1 ? smp[k].length * 8 : smp[k].byteLength;
  if (chainCount === 1) { for (const v of smp.mu) meanMu += v; meanMu /= smp.mu.length; }
  else { const a = smp.mu; for (let i = 0; i < a.length; i += 997) meanMu += a[i]; meanMu /= Math.ceil(a.length / 997); }
  const li = s.info().launch;
  s.close();
  const t4 = now();
  const updates = chainCount * (burn + sample) * 2;
  return { chains: chainCount, n_obs: data.length, burn, sample, thin: thinBy, kept, ctor_ms: t1 - t0, burn_ms: t2 - t1, sample_ms: t3 - t2, close_ms: t4 - t3,
           total_ms: t4 - t0, bytes_copied: bytes, gb_copied: bytes / 1e9, updates, updates_per_s: updates / ((t4 - t0) * 1e-3),
           updates_per_s_excl_ctor: updates / ((t3 - t1) * 1e-3), mean_mu: meanMu, launch: li };
}

const out = { node: process.version, require_ms: t_req };
if (opt('translated-only', 0)) {
  // a closure the family recogniser does not know (the README model with its priors written the other way round): translated to HIP and
  // compiled with hiprtc -- or, in a process that finds it there, loaded from the on-disk code-object cache.  ONE chain, like the reference.
  const swapped = function (state, data) {
    var lp = 0;
    lp += ld.unif(state.sigma, 0, 100);
    lp += ld.norm(state.mu, 0, 100);
    for (var i = 0; i < data.length; i++) lp += ld.norm(data[i], state.mu, state.sigma);
    return lp;
  };
  const heights = [183, 192, 182, 183, 177, 185, 188, 188, 182, 185];
  const t0 = now();
  const s = new mcmc.AmwgSampler({ sigma: { type: 'real', lower: 0 }, mu: { type: 'real' } }, swapped, heights);
  const t1 = now();
  s.burn(1000);
  const smp = s.sample(5000);
  const t2 = now();
  out.translated = { model: s.model, ctor_ms: t1 - t0, burn_sample_ms: t2 - t1, draws: smp.mu.length };
  s.close();
  try { out.code_cache = require(path.join(__dirname, '..', 'bayes.js_amd', 'mcmc.js')).code_cache_stats(); } catch (e) { out.code_cache = null; }
  console.log(JSON.stringify(out));
  process.exit(0);
}
out.many_chains = run(makeData(N), chains, nBurn, nSample, thin);
if (opt('translated-many', 1)) out.many_chains_translated = run(makeData(N), chains, nBurn, nSample, thin, true);
if (opt('single', 1)) {
  // the reference's own use (README.md:18-43): ONE chain on the ten heights, 1000 + 5000 steps; twice -- the second construction of the
  // same closure + geometry finds the compiled code object (in this process: in memory; in a NEW process: in the on-disk cache)
  const heights = [183, 192, 182, 183, 177, 185, 188, 188, 182, 185];
  out.single_chain_first = run(heights, 1, 1000, 5000, 1);
  out.single_chain_again = run(heights, 1, 1000, 5000, 1);
}
try { out.code_cache = require(path.join(__dirname, '..', 'bayes.js_amd', 'mcmc.js')).code_cache_stats(); } catch (e) { out.code_cache = null; }
console.log(JSON.stringify(out));
