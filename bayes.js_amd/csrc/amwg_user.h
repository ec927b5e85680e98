// amwg_user.h -- support code for translated closures (bayes.js_amd/translate.js emits calls to
// these).  Everything follows the reference's expression trees; see amwg_ld.h for the densities.
#pragma once
#include "amwg_div.h"
#include "amwg_ld.h"
#include "amwg_pass.h"
#include "amwg_twoval.h"
#include "amwg_kval.h"
#include "amwg_types.h"

namespace amwg {

// data array J of the closure: the first kInlineUserArrays pointers travel in the kernel arguments, further ones in a device table
template <int J>
AMWG_HD const void *user_arr(const DataRef &d) {
  if constexpr (J < kInlineUserArrays) return d.arr[J];
  else return d.arr_ext[J - kInlineUserArrays];
}

// Loop-invariant part of ld.norm(x, mean, sd) for a loop in which sd does not change
// (distributions.js:119-121): c = -0.5*log(2*pi) - log(sd), den = 2*sd*sd -- the same roundings,
// in the same order, as the full expression -- plus the double-double reciprocal of amwg_div.h.
struct NormInv { double c, den; Reciprocal y; bool fast; };
AMWG_HD NormInv norm_inv(double sd) {
  NormInv k;
  k.c = norm_c(-0.5 * log_v8(2 * kPi), sd);
  k.den = norm_den(sd);
  k.y = make_reciprocal(k.den);
  k.fast = mid_range(k.den);
  return k;
}
AMWG_HD double ld_norm_inv(double x, double mean, const NormInv &k) {
  const double t = x - mean;
  const double tt = t * t;
  // the quotient is the correctly rounded tt/den either way (amwg_div.h); '/' when a range precondition fails
  const double q = (k.fast && wide_range(tt)) ? div_by_invariant(tt, k.den, k.y) : tt / k.den;
  return k.c - q;
}

// Straight-line variants for a lane-split loop: the fast form always takes the 4-operation quotient
// and only RECORDS the exponent range of the numerators it saw (two 32-bit min/max per term, no
// branch); after the loop the generated code checks the range and the divisor and, if a precondition
// of amwg_div.h failed anywhere, restores the accumulator and runs the loop again in the slow form
// (IEEE division).  Both forms return the correctly rounded quotient, so the sum is the same bits.
AMWG_HD double ld_norm_fast(double x, double mean, const NormInv &k, uint32_t &rlo, uint32_t &rhi) {
  const double t = x - mean;
  const double tt = t * t;                 // >= +0: the sign bit is clear, hi_word orders like the magnitude
  const uint32_t h = (uint32_t)hi_word(tt);
  rlo = h < rlo ? h : rlo;
  rhi = h > rhi ? h : rhi;
  return k.c - div_by_invariant(tt, k.den, k.y);
}
AMWG_HD double ld_norm_slow(double x, double mean, const NormInv &k) {
  const double t = x - mean;
  return k.c - (t * t) / k.den;
}
AMWG_HD bool norm_range_ok(uint32_t rlo, uint32_t rhi) { return rlo >= 0x1A700000u && rhi <= 0x65700000u; }   // 2^-600 .. 2^600, zero excluded

// `for (i = 0; i < n; i++) lp += ld.norm(x[i], mean, sd)` over a whole f64 data array with loop-invariant mean and sd -- the likelihood
// loop of the README's model and of most closures -- compiles to the hand-scheduled pass of amwg_pass.h, the one the built-in Normal
// family runs: with G lanes per chain the staged pass over the array in LDS (x_staged), with ONE lane per chain the scalar-load pass over
// the array in global memory (x_global: the observations are wave-uniform).  Same operations in the same order as the closure's own loop
// (t = x - mean; t*t / (2*sd*sd); c - q; lp += term), the quotient by the 4-operation form of amwg_div.h when its range preconditions hold
// (divisor, mean and -- checked by the translator on the host -- every data value inside 2^-200..2^200 or zero) and by IEEE '/' otherwise.
template <class T, class U> struct SameType { static constexpr bool value = false; };
template <class T> struct SameType<T, T> { static constexpr bool value = true; };
// XT = double, or the u8 / i32 storage the translator picks for all-integer data (conversions are exact; the scalar-load pass is for doubles)
template <int G, class XT>
AMWG_HD double norm_data_loop(const XT *x_staged, const XT *x_global, int n, double mean, const NormInv &k, bool data_mid_range, int sub, double acc) {
  if (k.fast && data_mid_range && (mean == 0 || mid_range(__builtin_fabs(mean)))) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (G == 1 && SameType<XT, double>::value) return norm_pass_uniform<8>(x_global, mean, k.c, k.den, k.y, n, acc);
    else return norm_pass_staged<G, 8, false, XT>(x_staged, nullptr, StateView{nullptr}, mean, k.c, k.den, k.y, n, sub, acc);
#else
    for (int i = sub; i < n; i += G) { const double t = (double)x_global[i] - mean; acc += k.c - div_by_invariant(t * t, k.den, k.y); }
    return acc;
#endif
  }
  const XT *x = (G == 1 && SameType<XT, double>::value) ? x_global : x_staged;
  for (int i = sub; i < n; i += G) { const double t = (double)x[i] - mean; acc += k.c - (t * t) / k.den; }
  return acc;
}

// CERTIFIED TAIL of a translated closure (translate.js tailPlan; amwg_kernel.h "certified decisions"; the hand-written twin is NormalModel::log_post_approx).  A
// closure whose LAST statement is that loop -- lp = head; for (i) lp += ld.norm(x[i], mean, sd), mean and sd expressions of the state alone, x a whole f64 array --
// evaluates, as a real number, to  head + n c - S2 / den  with S2 = sum (x_i - mean)^2: two operations per observation where the expression's term takes eight.  The
// generated model hands the stepper that value with the bound the Normal family derives (amwg_models.h; `head` stands where its prior stands: the same fp64 number on
// both sides) -- eps = (2 n + 64) u (|head| + n |c| + 2 Q) 1.25 -- and the step kernel decides from it exactly as it does for the family: the expression (the closure's
// own body, one lane per chain = the reference's order) where a uniform falls inside the bound and wherever a value is stored.  One lane per chain only; the pass is
// the wavefront's (norm_sq_pass_wave: workgroups of up to 256 threads), the scalar-path pass in larger workgroups.
struct TailApprox { double value, eps; };
#if defined(__HIPCC__) || defined(__HIPCC_RTC__)      // (device code only: the host build of a generated model -- tests/host -- evaluates the closure itself)
template <class M, int G, int BT>
__device__ __forceinline__ TailApprox norm_tail_approx(const StateView &S, const DataRef &d, const unsigned char *smem, int sub) {
  static_assert(G == 1, "the certified tail of a translated closure is the one-lane one");
#if defined(__HIP_DEVICE_COMPILE__)
  const double head = M::template tail_head<1>(S, d, smem, 0);
  const NormInv k = norm_inv(M::tail_sd(S, d));
  const double mean = M::tail_mean(S, d);
  double S2;
  if constexpr (BT <= 256) S2 = norm_sq_pass_wave<kWaveBlock>(M::tail_x(d, smem), mean, M::kTailN, wave_scratch_of(d));      // (every lane of the wavefront takes part: the stepper's call site is wave-uniform)
  else S2 = norm_sq_pass_uniform<8>(M::tail_x_global(d), mean, M::kTailN);
  const double n = (double)M::kTailN;
  const double Q = S2 * k.y.hi, nc = n * k.c;
  const double mag = __builtin_fabs(head) + __builtin_fabs(nc) + 2.0 * Q;
  return TailApprox{(head + nc) - Q, (2.0 * n + 64.0) * 0x1p-53 * mag * 1.25};
#else
  (void)S; (void)d; (void)smem; (void)sub;
  return TailApprox{0.0, __builtin_inf()};
#endif
}
#endif

// The same for a GATHERED mean -- `for (i...) lp += ld.norm(y[i], state.theta[g[i]], sd)`, the likelihood loop of a model with group
// means (random effects): g a data array of small integers (stored as bytes), theta a parameter vector at state offset `base` with
// `n_groups` entries that g can reach.  PERIODIC (worked out by the translator for this lane count): g[i] == g[i % G] for every i, so a
// lane meets ONE group and reads its mean once (constant-mean pass); otherwise the pass that reads the indices two blocks and the means
// one block ahead.  Fast form only while every reachable mean is inside the range amwg_div.h needs (agreed on by a ballot of the chain's
// lanes); IEEE '/' otherwise.  Same operations in the same order as the closure's loop.
template <int G, bool PERIODIC, class XT>
AMWG_HD double norm_data_loop_gather(const XT *x, const uint8_t *g, const StateView &S, int base, int n_groups, int n, const NormInv &k,
                                     bool data_mid_range, int sub, double acc) {
  bool ok = k.fast && data_mid_range;
#if defined(__HIP_DEVICE_COMPILE__)
  {
    constexpr int L = G < 64 ? G : 64;
    const int lane = (int)(threadIdx.x & 63u);
    bool mine = true;
    for (int j = lane & (L - 1); j < n_groups; j += L) { const double th = S(base + j); mine = mine && (th == 0 || mid_range(__builtin_fabs(th))); }
    if constexpr (L == 1) ok = ok && mine;
    else {
      const uint64_t all = __ballot(mine);
      const uint64_t group = (L == 64 ? ~0ull : ((1ull << (L & 63)) - 1ull)) << (lane & ~(L - 1));
      ok = ok && ((all & group) == group);
    }
  }
  if (ok) {
    const StateView T{S.base + base};
    if constexpr (PERIODIC) {
      const double m = sub < n ? T(g[sub]) : 0.0;
      return norm_pass_staged<G, 8, false, XT>(x, nullptr, T, m, k.c, k.den, k.y, n, sub, acc);
    } else {
      return norm_pass_staged<G, 4, true, XT>(x, g, T, 0.0, k.c, k.den, k.y, n, sub, acc);
    }
  }
#else
  for (int j = 0; j < n_groups; ++j) { const double th = S(base + j); ok = ok && (th == 0 || mid_range(__builtin_fabs(th))); }
  if (ok) {
    for (int i = sub; i < n; i += G) { const double t = (double)x[i] - S(base + g[i]); acc += k.c - div_by_invariant(t * t, k.den, k.y); }
    return acc;
  }
#endif
  for (int i = sub; i < n; i += G) { const double t = (double)x[i] - S(base + g[i]); acc += k.c - (t * t) / k.den; }
  return acc;
}

// ld.bern(x, p) for a loop in which p does not change (distributions.js:228-230): the two values
// log(1*p + 0*(1-p)) and log(0*p + 1*(1-p)) the expression can take, selected per observation.
struct BernInv { double l1, l0; };
AMWG_HD BernInv bern_inv(double p) { return BernInv{ld_bern(1.0, p), ld_bern(0.0, p)}; }
AMWG_HD double ld_bern_inv(double x, const BernInv &k) { return x == 1 ? k.l1 : (x == 0 ? k.l0 : -kInf); }
// same, for a data array the translator has checked to hold only 0s and 1s
template <class T> AMWG_HD double ld_bern_inv01(T x, const BernInv &k) { return x != 0 ? k.l1 : k.l0; }

// ld.pois / ld.binom whose data-only part (lfactorial(x) resp. lchoose(size, x), distributions.js:79-86)
// was evaluated once per observation on the host by the same formula
AMWG_HD double ld_pois_pre(double x, double lambda, double lfact_x) { return x < 0 ? -kInf : log_v8(lambda) * x - lambda - lfact_x; }
// the same under a log link written in place -- ld.pois(y[i], Math.exp(eta)): exp and the log of its value share their argument reduction
// (amwg_math.h exp_log_v8: same bits as log_v8(exp_v8(eta)))
AMWG_HD double ld_pois_pre_exp(double x, double eta, double lfact_x) {
  double lambda;
  const double lg = exp_log_v8(eta, lambda, ExpLogLiterals{});
  return x < 0 ? -kInf : lg * x - lambda - lfact_x;
}
AMWG_HD double ld_pois_exp(double x, double eta) {
  double lambda;
  const double lg = exp_log_v8(eta, lambda, ExpLogLiterals{});
  return x < 0 ? -kInf : lg * x - lambda - lfactorial_js(x);
}
AMWG_HD double ld_binom_pre(double x, double size, double prob, double lchoose_size_x) {
  if (x > size || x < 0) return -kInf;
  if (prob == 0 || prob == 1) return (size * prob) == x ? 0.0 : -kInf;
  return lchoose_size_x + x * log_v8(prob) + (size - x) * log_v8(1 - prob);
}

// `for (i = 0; i < n; i++) lp += ld.bern(x[i], p)` over a 0/1 data array with one lane per chain: the six tables of
// amwg_twoval.h for that array (built by the translator, stored back to back as 32-bit words) and the exact fast-forward
AMWG_HD double bern_loop_one_lane(double acc, const BernInv &k, const int32_t *tab, int n) {
  const size_t W = two_valued_words(n);
  const uint32_t *t = reinterpret_cast<const uint32_t *>(tab);
  const BitData B{t, t + W, t + 2 * W, t + 3 * W, t + 4 * W, t + 5 * W, n};
  return two_valued_sum(acc, k.l1, k.l0, B);
}

// `for (i = 0; i < n; i++) lp += ld.pois(y[i], rate)` (or ld.binom with loop-invariant size / prob) over a data array with K <= 16 distinct
// values, one lane per chain: c[k] = the term of the k-th distinct value (computed by the generated code with the same function the
// term-by-term loop calls), the tables of amwg_kval.h for that array (built by the translator) and the exact fast-forward
template <int K>
AMWG_HD double kval_loop_one_lane(double acc, const double (&c)[K], const int32_t *tab, const uint8_t *idx, int n) {
  const KValData B{reinterpret_cast<const uint32_t *>(tab), idx, n};
  return k_valued_sum<K>(acc, c, B);
}

// JavaScript operators that differ from C++
// `%` on numbers (ECMA-262 6.1.6.1.6): the exact remainder of the truncating division, with the sign of the DIVIDEND -- also when
// the result is zero (-0 % 3 is -0, -6 % 3 is -0); NaN for a NaN operand, an infinite dividend or a zero divisor; the dividend
// itself for an infinite divisor.  Written out on the bit patterns (restoring shift-subtract division of the significands, one
// quotient bit per exponent step) instead of __builtin_fmod: that lowers to a toolchain-specific `frem` expansion / ocml call on
// the device, whose zero results lost the sign on the GPU box in round 1, and to glibc on the host.  Same code on both here.
AMWG_HD double js_mod(double a, double b) {
  const uint64_t ua = f64_bits(a), ub = f64_bits(b);
  const uint64_t sign = ua & 0x8000000000000000ull;
  uint64_t ma = ua & 0x7fffffffffffffffull, mb = ub & 0x7fffffffffffffffull;
  if (ma >= 0x7ff0000000000000ull || mb > 0x7ff0000000000000ull || mb == 0) return __builtin_nan("");
  if (ma < mb) return a;                      // includes b = +-inf and a = +-0
  if (ma == mb) return bits_f64(sign);        // +-0 with the dividend's sign
  int ea = (int)(ma >> 52), eb = (int)(mb >> 52);
  // significands with the hidden bit at position 52 (subnormals are shifted up, exponent counted down)
  if (ea == 0) { const int sh = __builtin_clzll(ma) - 11; ma <<= sh; ea = 1 - sh; }
  else ma = (ma & 0x000fffffffffffffull) | 0x0010000000000000ull;
  if (eb == 0) { const int sh = __builtin_clzll(mb) - 11; mb <<= sh; eb = 1 - sh; }
  else mb = (mb & 0x000fffffffffffffull) | 0x0010000000000000ull;
  for (; ea > eb; --ea) {
    if (ma >= mb) ma -= mb;
    ma <<= 1;
  }
  if (ma >= mb) ma -= mb;
  if (ma == 0) return bits_f64(sign);
  const int sh = __builtin_clzll(ma) - 11;    // renormalise; the remainder is < b so it fits
  ma <<= sh;
  ea -= sh;
  if (ea > 0) ma = (ma & 0x000fffffffffffffull) | ((uint64_t)ea << 52);
  else ma >>= (1 - ea);                       // subnormal result (exact: the low bits shifted out are zero)
  return bits_f64(ma | sign);
}
AMWG_HD double js_max(double a, double b) {   // Math.max: NaN if either is NaN, +0 > -0
  if (a != a || b != b) return __builtin_nan("");
  if (a == b) return (a == 0 && __builtin_signbit(a)) ? b : a;
  return a > b ? a : b;
}
AMWG_HD double js_min(double a, double b) {
  if (a != a || b != b) return __builtin_nan("");
  if (a == b) return (a == 0 && __builtin_signbit(a)) ? a : b;
  return a < b ? a : b;
}
AMWG_HD double js_sign(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : x); }
AMWG_HD double js_trunc(double x) { return __builtin_trunc(x); }
// ToInt32 (ECMA-262 7.1.6): truncate, reduce modulo 2^32 into [-2^31, 2^31); NaN and infinities give 0.  Bitwise operators, Math.imul and
// Math.clz32 work on this value (`x | 0`, `~~x`, `x >>> 0` are the usual integer-truncation idioms).
AMWG_HD int32_t js_toint32(double x) {
  if (!(__builtin_fabs(x) < __builtin_inf())) return 0;
  const double t = __builtin_trunc(x);
  if (t >= -2147483648.0 && t <= 2147483647.0) return (int32_t)t;
  double m = js_mod(t, 4294967296.0);                   // exact; sign of t
  if (m < 0) m += 4294967296.0;
  return (int32_t)(uint32_t)m;
}
AMWG_HD double js_bitor(double a, double b) { return (double)(js_toint32(a) | js_toint32(b)); }
AMWG_HD double js_bitand(double a, double b) { return (double)(js_toint32(a) & js_toint32(b)); }
AMWG_HD double js_bitxor(double a, double b) { return (double)(js_toint32(a) ^ js_toint32(b)); }
AMWG_HD double js_bitnot(double a) { return (double)(~js_toint32(a)); }
AMWG_HD double js_shl(double a, double b) { return (double)(int32_t)((uint32_t)js_toint32(a) << ((uint32_t)js_toint32(b) & 31u)); }
AMWG_HD double js_shr(double a, double b) { return (double)(js_toint32(a) >> ((uint32_t)js_toint32(b) & 31u)); }
AMWG_HD double js_ushr(double a, double b) { return (double)((uint32_t)js_toint32(a) >> ((uint32_t)js_toint32(b) & 31u)); }
AMWG_HD double js_imul(double a, double b) { return (double)(int32_t)((uint32_t)js_toint32(a) * (uint32_t)js_toint32(b)); }
AMWG_HD double js_clz32(double a) { const uint32_t u = (uint32_t)js_toint32(a); return u == 0 ? 32.0 : (double)__builtin_clz(u); }
AMWG_HD double js_fround(double a) { return (double)(float)a; }

}  // namespace amwg

#include "amwg_rows.h"      // lane-local re-evaluation + sweep prefetch for closures with a row plan (device only)
#include "amwg_ptail.h"     // certified values for closures that end in a log-link Poisson loop (device only)
