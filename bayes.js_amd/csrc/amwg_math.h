// amwg_math.h -- fp64 exp/log for the AMWG kernel, bit-identical to the JavaScript
// engine the reference runs on.
//
// The reference computes every transcendental with V8's Math.exp / Math.log
// (mcmc.js:51, 527, 578; distributions.js:94-95), which are the Sun fdlibm algorithms
// (V8 src/base/ieee754.cc).  To make "same seed => same accept decisions, same draws"
// a testable statement rather than a statistical one, the kernel evaluates the same
// published algorithms operation for operation (compile with -ffp-contract=off; the only
// fused operations are the explicit fma() calls of amwg_div.h).  tests/test_core_math.py
// pins the host build against 120 000 outputs of Node's Math.exp/Math.log and the device
// build against the host build.
//
// The algorithms restated here are those of FreeBSD msun / Sun fdlibm as V8 carries them (src/base/ieee754.cc), whose files bear:
//   ====================================================
//   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
//   Developed at SunSoft, a Sun Microsystems, Inc. business.
//   Permission to use, copy, modify, and distribute this
//   software is freely granted, provided that this notice
//   is preserved.
//   ====================================================
// (e_exp.c, e_log.c, e_pow.c, s_log1p.c, s_expm1.c, s_tanh.c, s_atan.c, e_log10.c; for amwg_trig.h also k_rem_pio2.c, e_rem_pio2.c, k_sin.c,
// k_cos.c, k_tan.c, e_asin.c, e_acos.c, e_atan2.c, e_sinh.c, e_cosh.c, s_asinh.c, e_acosh.c, e_atanh.c, s_cbrt.c, e_log2.c.)
//
// GPU shaping: both functions are straight-line on the common path; range/special handling
// is folded into selects or rare branches so 64 chains with different arguments stay converged.
#pragma once
#include "amwg_stdint.h"

#if defined(__HIPCC__)
#define AMWG_HD __host__ __device__ __forceinline__
#define AMWG_HD_OUTLINE __host__ __device__ inline __attribute__((noinline))   // large, rarely executed bodies: one copy per kernel
#else
#define AMWG_HD inline
#define AMWG_HD_OUTLINE inline
#endif

namespace amwg {

AMWG_HD uint64_t f64_bits(double x) { return __builtin_bit_cast(uint64_t, x); }
AMWG_HD double bits_f64(uint64_t u) { return __builtin_bit_cast(double, u); }
AMWG_HD int32_t hi_word(double x) { return (int32_t)(f64_bits(x) >> 32); }
AMWG_HD uint32_t lo_word(double x) { return (uint32_t)f64_bits(x); }
AMWG_HD double set_hi_word(double x, int32_t hi) {
  return bits_f64(((uint64_t)(uint32_t)hi << 32) | (uint64_t)lo_word(x));
}

// exp(x) = 2^k * exp(r), r = x - k ln2 in two pieces, exp(r) = 1 + r + r*c/(2-c),
// c = r - r^2 * P(r^2).
// Full fdlibm control flow (every special case); the kernel reaches it only on the rare path.
AMWG_HD double exp_v8_full(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double inv_ln2 = 1.44269504088896338700e+00, two_m1000 = 9.33263618503218878990e-302;
  const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
               P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  const uint32_t hx_signed = (uint32_t)hi_word(x);
  const bool neg = (hx_signed >> 31) != 0;
  const uint32_t hx = hx_signed & 0x7fffffffu;

  if (hx >= 0x40862E42u) {  // |x| >= 709.78, inf or NaN: rare
    if (hx >= 0x7ff00000u) {
      if (((hx & 0xfffffu) | lo_word(x)) != 0) return x + x;
      return neg ? 0.0 : x;
    }
    if (x > 7.09782712893383973096e+02) return 1.0e+300 * 1.0e+300;
    if (x < -7.45133219101941108420e+02) return two_m1000 * two_m1000;
  }
  double hi = 0.0, lo = 0.0;
  int32_t k = 0;
  if (hx > 0x3fd62e42u) {  // |x| > 0.5 ln2
    if (hx < 0x3FF0A2B2u) {  // |x| < 1.5 ln2
      if (x == 1.0) return 2.718281828459045;  // V8 returns Math.E here
      hi = neg ? x + ln2_hi : x - ln2_hi;      // x - (-ln2_hi) == x + ln2_hi exactly
      lo = neg ? -ln2_lo : ln2_lo;
      k = neg ? -1 : 1;
    } else {
      k = (int32_t)(inv_ln2 * x + (neg ? -0.5 : 0.5));
      const double t = (double)k;
      hi = x - t * ln2_hi;
      lo = t * ln2_lo;
    }
    x = hi - lo;
  } else if (hx < 0x3e300000u) {  // |x| < 2^-28
    return 1.0 + x;               // fdlibm's `huge + x > one` guard is always true here
  }
  const double t = x * x;
  const double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
  const double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
  if (k >= -1021) {
    if (k == 1024) return y * 2.0 * 8.98846567431157953865e+307;
    return set_hi_word(y, hi_word(y) + (k << 20));
  }
  return set_hi_word(y, hi_word(y) + ((k + 1000) << 20)) * two_m1000;
}

// log(x): x = 2^k (1+f), sqrt(2)/2 < 1+f < sqrt(2); s = f/(2+f); log(1+f) = f - s (f - R(s^2)).
// Full fdlibm control flow; rare path of log_v8 below.
AMWG_HD double log_v8_full(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double two54 = 1.80143985094819840000e+16;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  int32_t hx = hi_word(x), k = 0;
  if (hx < 0x00100000) {  // zero, negative, subnormal: rare
    if (((hx & 0x7fffffff) | (int32_t)lo_word(x)) == 0) return -two54 / 0.0;
    if (hx < 0) return (x - x) / 0.0;
    k -= 54;
    x *= two54;
    hx = hi_word(x);
  }
  if (hx >= 0x7ff00000) return x + x;
  k += (hx >> 20) - 1023;
  hx &= 0x000fffff;
  const int32_t i = (hx + 0x95f64) & 0x100000;
  x = set_hi_word(x, hx | (i ^ 0x3ff00000));
  k += (i >> 20);
  const double f = x - 1.0;
  const double dk = (double)k;
  if ((0x000fffff & (2 + hx)) < 3) {  // |f| < 2^-20: rare
    if (f == 0.0) return (k == 0) ? 0.0 : dk * ln2_hi + dk * ln2_lo;
    const double R = f * f * (0.5 - 0.33333333333333333 * f);
    return (k == 0) ? f - R : dk * ln2_hi - ((R - dk * ln2_lo) - f);
  }
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  const double R = t2 + t1;
  if (((hx - 0x6147a) | (0x6b851 - hx)) > 0) {
    const double hfsq = 0.5 * f * f;
    return (k == 0) ? f - (hfsq - s * (hfsq + R)) : dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  }
  return (k == 0) ? f - s * (f - R) : dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

// ---- straight-line front ends -------------------------------------------------------------------
// Same values as the *_full versions (tests pin both against V8), shaped for a 64-lane SIMD: the
// three argument-reduction cases of exp and the two polynomial tails of log become selects, the
// k == 0 special forms are folded into the general ones (they are the general formula with
// lo = 0, hi = x resp. dk = 0: x/(c-2) == -(x/(2-c)), 0 - a == -a, a + 0 == a, all exact), and
// everything else (|x| >= 708, |x| < 2^-28, exp(1); log of <= 0, subnormal, inf/NaN, |f| < 2^-20)
// leaves through ONE rarely-taken branch.
//
// quot_plain(a, b): a / b for operands that need none of the exponent juggling of the general fp64 division -- b within
// [1, 4), a zero or of magnitude in [2^-900, 2^100].  On gfx950 `/` expands to v_div_scale x2, v_rcp_f64, two Newton
// steps, a quotient, a residual, v_div_fmas and v_div_fixup; for such operands the two v_div_scale are the identity and
// v_div_fixup returns its first operand, so the same reciprocal / Newton / residual instructions without them give the
// same bits (tests/test_gpu_math.py compares with `/` on the device).  The host build divides.
AMWG_HD double quot_plain(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  const double q = a * y;
  const double r = __builtin_fma(-b, q, a);
  return __builtin_fma(r, y, q);
#else
  return a / b;
#endif
}

// The constants of exp / log as a type: ExpLogLiterals are compile-time values (scalar registers or literals once inlined);
// ExpLogRegs (below) holds the same numbers in per-lane registers for a loop whose scalar registers are better spent on pointers.
struct ExpLogLiterals {
  static constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, inv_ln2 = 1.44269504088896338700e+00;
  static constexpr double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                          P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  static constexpr double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                          Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                          Lg7 = 1.479819860511658591e-01;
};
struct ExpLogRegs {
  double ln2_hi, ln2_lo, inv_ln2, P1, P2, P3, P4, P5, Lg1, Lg2, Lg3, Lg4, Lg5, Lg6, Lg7;
};
AMWG_HD ExpLogRegs exp_log_regs() {
  typedef ExpLogLiterals L;
  ExpLogRegs k{L::ln2_hi, L::ln2_lo, L::inv_ln2, L::P1, L::P2, L::P3, L::P4, L::P5, L::Lg1, L::Lg2, L::Lg3, L::Lg4, L::Lg5, L::Lg6, L::Lg7};
#if defined(__HIP_DEVICE_COMPILE__)
  // "+v": the value now lives in a vector register as far as the compiler can tell (it would otherwise rematerialise the literal into
  // a scalar pair wherever it is used)
  asm volatile("" : "+v"(k.ln2_hi), "+v"(k.ln2_lo), "+v"(k.inv_ln2), "+v"(k.P1), "+v"(k.P2), "+v"(k.P3), "+v"(k.P4), "+v"(k.P5));
  // (Lg1..Lg7 stay compile-time values: with all fifteen in vector registers the Poisson kernel needs 263 of them, a wave per SIMD less)
#endif
  return k;
}

// rare arguments of exp: |x| >= ~708 (overflow / underflow / denormal scaling / inf / NaN), |x| < 2^-28, exactly 1.0 (V8 returns Math.E),
// and the sliver 0x3fd62e42_00000000 <= |x| <= 0x3fd62e42_ffffffff around ln2/2 (0x3fd62e42_fefa39ef), the only place where fdlibm's
// THREE-way choice of k (0 below 0.5 ln2 by the high word; +-1 up to 1.5 ln2; (int)(x/ln2 +- 0.5) beyond) differs from the last formula
// applied everywhere: below the sliver x/ln2 < 0.5 - 3e-7, so the formula gives 0; above it and below 1.5 ln2 (high word < 0x3ff0a2b2,
// i.e. 2e-7 short of it) x/ln2 +- 0.5 lies within (1 + 1e-9, 2 - 3e-7) in magnitude, so it gives +-1 (tests/host/explog_fuzz.cpp walks
// both edges).
AMWG_HD bool exp_is_rare(double x) {
  const uint32_t hx = (uint32_t)hi_word(x) & 0x7fffffffu;
  return hx >= 0x40862000u || hx < 0x3e300000u || x == 1.0 || hx == 0x3fd62e42u;
}

// x = t ln2 + r (t an integer-valued double with the sign of x, possibly -0); t_hi = t*ln2_hi and lo = t*ln2_lo as fdlibm forms them;
// y = exp(r) in (0.70, 1.42); exp(x) = y * 2^k.  Not for the arguments exp_is_rare() names.
struct ExpParts { double t, t_hi, lo, y; int32_t k; };
template <class K>
AMWG_HD ExpParts exp_parts(double x, const K &c) {
  ExpParts e;
  // (int)(x/ln2 +- 0.5) of fdlibm, sign-symmetric: truncate |x/ln2| + 0.5 and put the sign back (two selects and a compare fewer).
  // t = -0 for a negative x with k = 0: t_hi = -0, lo = -0, and x - (-0), hi - (-0), (-0) - q (q != 0) are what +0 gives.
  e.t = __builtin_copysign(__builtin_trunc(__builtin_fabs(c.inv_ln2 * x) + 0.5), x);
  e.k = (int32_t)e.t;
  e.t_hi = e.t * c.ln2_hi;
  const double hi = x - e.t_hi;
  e.lo = e.t * c.ln2_lo;
  const double r = hi - e.lo;
  const double rr = r * r;
  const double cc = r - rr * (c.P1 + rr * (c.P2 + rr * (c.P3 + rr * (c.P4 + rr * c.P5))));
  // 2 - cc in (1.6, 2.4); r*cc is 0 or >= 2^-150 in magnitude (|r| >= ulp(ln2-multiple) of a |x| >= 2^-28)
  e.y = 1.0 - ((e.lo - quot_plain(r * cc, 2.0 - cc)) - hi);
  return e;
}

AMWG_HD double exp_v8(double x) {
  if (__builtin_expect(exp_is_rare(x), 0)) return exp_v8_full(x);
  const ExpParts e = exp_parts(x, ExpLogLiterals{});
  return set_hi_word(e.y, hi_word(e.y) + (e.k << 20));
}

AMWG_HD double log_v8(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  const int32_t hx0 = hi_word(x);
  const int32_t hx = hx0 & 0x000fffff;
  // rare: x <= 0 or subnormal, inf/NaN, or |f| < 2^-20 (x within 2^-20 of a power of two)
  if (__builtin_expect(hx0 < 0x00100000 || hx0 >= 0x7ff00000 || (0x000fffff & (2 + hx)) < 3, 0)) return log_v8_full(x);
  const int32_t i = (hx + 0x95f64) & 0x100000;
  const double m = set_hi_word(x, hx | (i ^ 0x3ff00000));
  const int32_t k = (hx0 >> 20) - 1023 + (i >> 20);
  const double f = m - 1.0;
  const double dk = (double)k;
  const double s = quot_plain(f, 2.0 + f);      // |f| in [2^-20, 0.42), 2 + f in (1.7, 2.42)
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double a = dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  const double b = dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
  return (((hx - 0x6147a) | (0x6b851 - hx)) > 0) ? a : b;
}

// Out-of-line copies for call sites that are rarely reached from inside the step loop (the rejection test of rnorm, a batch boundary, a
// changed sd): inlined there, their bodies and two dozen polynomial constants would sit in the hot loop's code and scalar registers.
#if defined(AMWG_X_NOCOLD)
AMWG_HD double log_v8_cold(double x) { return log_v8(x); }
AMWG_HD double exp_v8_cold(double x) { return exp_v8(x); }
#else
AMWG_HD_OUTLINE double log_v8_cold(double x) { return log_v8(x); }
AMWG_HD_OUTLINE double exp_v8_cold(double x) { return exp_v8(x); }
#endif

// lam = exp_v8(x) and log_v8(lam) in one go -- the Poisson log-density's  y*log(lambda) - lambda  with lambda = exp(eta)
// (distributions.js:282-284 under a log link).  log() starts by splitting its argument into 2^k (1 + f) with 1 + f in [sqrt(2)/2, sqrt(2)):
// for lam = y * 2^k straight out of exp() that split is (k, y) itself whenever y's high word lies in [0x3fe6a09c, 0x3ff6a09c) (fdlibm
// draws the line at significand high word 0x6a09c), which is everything but two slivers at |r| ~ ln2/2 -- so f = y - 1, dk = t, and the
// products dk*ln2_hi, dk*ln2_lo are the t_hi, lo exp() already formed (t = -0 where dk = +0: the terms they are added to are non-zero).
// The tail selection (fdlibm: 0x6147a <= significand high word <= 0x6b851 takes the form with f*f/2) and the |f| < 2^-20 test are
// restated on tmp = hi(y) - 0x3fe6a09c.  Anything else goes through exp_v8 / log_v8 as they are.  Same bits as log_v8(exp_v8(x)).
template <class K>
AMWG_HD double exp_log_v8(double x, double &lam, const K &c) {
  if (__builtin_expect(exp_is_rare(x), 0)) {
    lam = exp_v8_cold(x);
    return log_v8_cold(lam);
  }
  const ExpParts e = exp_parts(x, c);
  const int32_t hy = hi_word(e.y);
  lam = set_hi_word(e.y, hy + (e.k << 20));
  constexpr uint32_t base = 0x3fe6a09cu;
  const uint32_t tmp = (uint32_t)hy - base;
  // not the plain split, or |f| < 2^-20 (hi(y) in {0x3feffffe, 0x3fefffff, 0x3ff00000})
  if (__builtin_expect(tmp >= 0x100000u || (tmp - (0x3feffffeu - base)) < 3u, 0)) return log_v8_cold(lam);
  const double f = e.y - 1.0;
  const double s = quot_plain(f, 2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * (c.Lg2 + w * (c.Lg4 + w * c.Lg6));
  const double t2 = z * (c.Lg1 + w * (c.Lg3 + w * (c.Lg5 + w * c.Lg7)));
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double a = e.t_hi - ((hfsq - (s * (hfsq + R) + e.lo)) - f);
  const double b = e.t_hi - ((s * (f - R) - e.lo) - f);
  // form b: significand high word outside [0x6147a, 0x6b851], i.e. hi(y) in [0x3fe6b852, 0x3ff61479]
  return (tmp - (0x3fe6b852u - base)) <= (0x3ff61479u - 0x3fe6b852u) ? b : a;
}

// exp_log_v8 for U arguments at once and without its branches: every step is taken for all U before the next one, so U independent
// dependent chains sit side by side in ONE basic block (the compiler keeps that order; two calls one after the other are scheduled one
// after the other).  The values are garbage -- but no trap -- where `rare` comes back set: the caller then goes through exp_v8_cold /
// log_v8_cold for those lanes.  Same operations as exp_parts + exp_log_v8 (tests/host/explog_fuzz.cpp compares this form too).
template <int U, class K>
AMWG_HD void exp_log_v8_open(const double (&x)[U], double (&lam)[U], double (&lg)[U], bool (&rare)[U], const K &c) {
#define AMWG_EACH for (int u = 0; u < U; ++u)
  double t[U], t_hi[U], hi[U], lo[U], r[U], rr[U], p[U], cc[U], num[U], den[U], q[U];
#if defined(__HIP_DEVICE_COMPILE__)
  double y[U], e[U], res[U];
#endif
  uint32_t tmp[U];
  constexpr uint32_t base = 0x3fe6a09cu;
#pragma unroll
  AMWG_EACH t[u] = __builtin_copysign(__builtin_trunc(__builtin_fabs(c.inv_ln2 * x[u]) + 0.5), x[u]);
#pragma unroll
  AMWG_EACH t_hi[u] = t[u] * c.ln2_hi;
#pragma unroll
  AMWG_EACH hi[u] = x[u] - t_hi[u];
#pragma unroll
  AMWG_EACH lo[u] = t[u] * c.ln2_lo;
#pragma unroll
  AMWG_EACH r[u] = hi[u] - lo[u];
#pragma unroll
  AMWG_EACH rr[u] = r[u] * r[u];
#pragma unroll
  AMWG_EACH p[u] = c.P4 + rr[u] * c.P5;
#pragma unroll
  AMWG_EACH p[u] = c.P3 + rr[u] * p[u];
#pragma unroll
  AMWG_EACH p[u] = c.P2 + rr[u] * p[u];
#pragma unroll
  AMWG_EACH p[u] = c.P1 + rr[u] * p[u];
#pragma unroll
  AMWG_EACH cc[u] = r[u] - rr[u] * p[u];
#pragma unroll
  AMWG_EACH { num[u] = r[u] * cc[u]; den[u] = 2.0 - cc[u]; }
  // quot_plain(num, den), step by step
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  AMWG_EACH y[u] = __builtin_amdgcn_rcp(den[u]);
#pragma unroll
  AMWG_EACH e[u] = __builtin_fma(-den[u], y[u], 1.0);
#pragma unroll
  AMWG_EACH y[u] = __builtin_fma(y[u], e[u], y[u]);
#pragma unroll
  AMWG_EACH e[u] = __builtin_fma(-den[u], y[u], 1.0);
#pragma unroll
  AMWG_EACH y[u] = __builtin_fma(y[u], e[u], y[u]);
#pragma unroll
  AMWG_EACH q[u] = num[u] * y[u];
#pragma unroll
  AMWG_EACH res[u] = __builtin_fma(-den[u], q[u], num[u]);
#pragma unroll
  AMWG_EACH q[u] = __builtin_fma(res[u], y[u], q[u]);
#else
  AMWG_EACH q[u] = num[u] / den[u];
#endif
  double ey[U], f[U], s[U], z[U], w[U], t1[U], t2[U], R[U], hfsq[U], a[U], b[U];
#pragma unroll
  AMWG_EACH ey[u] = 1.0 - ((lo[u] - q[u]) - hi[u]);
#pragma unroll
  AMWG_EACH {
    const int32_t hy = hi_word(ey[u]);
    lam[u] = set_hi_word(ey[u], hy + ((int32_t)t[u] << 20));
    tmp[u] = (uint32_t)hy - base;
    rare[u] = exp_is_rare(x[u]) || tmp[u] >= 0x100000u || (tmp[u] - (0x3feffffeu - base)) < 3u;
  }
#pragma unroll
  AMWG_EACH { f[u] = ey[u] - 1.0; den[u] = 2.0 + f[u]; }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  AMWG_EACH y[u] = __builtin_amdgcn_rcp(den[u]);
#pragma unroll
  AMWG_EACH e[u] = __builtin_fma(-den[u], y[u], 1.0);
#pragma unroll
  AMWG_EACH y[u] = __builtin_fma(y[u], e[u], y[u]);
#pragma unroll
  AMWG_EACH e[u] = __builtin_fma(-den[u], y[u], 1.0);
#pragma unroll
  AMWG_EACH y[u] = __builtin_fma(y[u], e[u], y[u]);
#pragma unroll
  AMWG_EACH s[u] = f[u] * y[u];
#pragma unroll
  AMWG_EACH res[u] = __builtin_fma(-den[u], s[u], f[u]);
#pragma unroll
  AMWG_EACH s[u] = __builtin_fma(res[u], y[u], s[u]);
#else
  AMWG_EACH s[u] = f[u] / den[u];
#endif
#pragma unroll
  AMWG_EACH z[u] = s[u] * s[u];
#pragma unroll
  AMWG_EACH w[u] = z[u] * z[u];
#pragma unroll
  AMWG_EACH { t1[u] = c.Lg4 + w[u] * c.Lg6; t2[u] = c.Lg5 + w[u] * c.Lg7; }
#pragma unroll
  AMWG_EACH { t1[u] = c.Lg2 + w[u] * t1[u]; t2[u] = c.Lg3 + w[u] * t2[u]; }
#pragma unroll
  AMWG_EACH { t1[u] = w[u] * t1[u]; t2[u] = c.Lg1 + w[u] * t2[u]; }
#pragma unroll
  AMWG_EACH { t2[u] = z[u] * t2[u]; hfsq[u] = 0.5 * f[u] * f[u]; }
#pragma unroll
  AMWG_EACH R[u] = t2[u] + t1[u];
#pragma unroll
  AMWG_EACH { a[u] = s[u] * (hfsq[u] + R[u]); b[u] = s[u] * (f[u] - R[u]); }
#pragma unroll
  AMWG_EACH { a[u] = hfsq[u] - (a[u] + lo[u]); b[u] = b[u] - lo[u]; }
#pragma unroll
  AMWG_EACH {
    const double sel = (tmp[u] - (0x3fe6b852u - base)) <= (0x3ff61479u - 0x3fe6b852u) ? b[u] : a[u];
    lg[u] = t_hi[u] - (sel - f[u]);
  }
#undef AMWG_EACH
}



// ---- pow(x, y): V8's Math.pow (src/base/ieee754.cc pow, the fdlibm e_pow.c algorithm; V8 groups the
// final quotient differently from fdlibm -- marked below -- and Node's values follow V8).  Used by
// ld.t, ld.weibull and user closures; Math.pow(t, 2) never gets here (it is exactly t*t).
// exp(x), |x| <= 700, to a relative error below 2^-46 -- NOT V8's exp (exp_v8 above is), for the certified pass of the Poisson family, which uses the value with
// its bound only (amwg_models.h PoisGlmModel::log_post_approx).  k = round(x / ln2); r = x - k ln2 by two fused steps (|r| <= 0.3466: the products are exact
// inside the fma, each step rounds a value of that size: 2 x 2^-53 x 0.35); a polynomial of degree 11 by Horner's rule in fmas -- the one that interpolates exp at
// the twelve Chebyshev nodes of [-ln2 / 2, ln2 / 2] (tools/exp_poly.py: 60-digit arithmetic; with its coefficients rounded to doubles it is 1.7e-17 from exp in exact
// arithmetic, c0 and c1 round to 1) --: roundings <= 22 x 2^-53 x e^|r| / e^-|r| = 4.9e-15 (the classical bound; measured 1.5e-16, tests/host/explog_fuzz.cpp); times
// 2^k, exact.  (Rounds 5 and 6 carried the Taylor polynomial of degree 13 -- the degree 11 one is 8.8e-15 off at r = -ln2 / 2: two fused steps per exponential that
// better coefficients make unnecessary.)
// 17 operations, none of them a reciprocal.  C: the coefficients as a type (literals, or per-lane registers: ExpTaylorRegs).
struct ExpTaylorLiterals {
  static constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, inv_ln2 = 1.44269504088896338700e+00;
  static constexpr double c2 = 0x1.0000000000011p-1, c3 = 0x1.555555555555ap-3, c4 = 0x1.555555554f0cep-5, c5 = 0x1.111111110f225p-7, c6 = 0x1.6c16c187fbe13p-10,
                          c7 = 0x1.a01a01b14379ap-13, c8 = 0x1.a01991ac8440bp-16, c9 = 0x1.71ddf5749b43dp-19, c10 = 0x1.28b40581fba5cp-22, c11 = 0x1.af631d03b18bfp-26;
};
struct ExpTaylorRegs { double ln2_hi, ln2_lo, inv_ln2, c2, c3, c4, c5, c6, c7, c8, c9, c10, c11; };
AMWG_HD ExpTaylorRegs exp_taylor_regs() {
  typedef ExpTaylorLiterals L;
  ExpTaylorRegs k{L::ln2_hi, L::ln2_lo, L::inv_ln2, L::c2, L::c3, L::c4, L::c5, L::c6, L::c7, L::c8, L::c9, L::c10, L::c11};
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(k.ln2_hi), "+v"(k.ln2_lo), "+v"(k.inv_ln2), "+v"(k.c2), "+v"(k.c3), "+v"(k.c4), "+v"(k.c5), "+v"(k.c6), "+v"(k.c7), "+v"(k.c8), "+v"(k.c9),
               "+v"(k.c10), "+v"(k.c11));      // (vector registers: a 64-bit literal cannot be an operand of v_fma_f64, and the scalar ones are few)
#endif
  return k;
}
template <class C>
AMWG_HD double exp_bounded(double x, const C &c) {
  const double t = __builtin_rint(x * c.inv_ln2);
  double r = __builtin_fma(-t, c.ln2_hi, x);
  r = __builtin_fma(-t, c.ln2_lo, r);
  double p = __builtin_fma(c.c11, r, c.c10);
  p = __builtin_fma(p, r, c.c9);
  p = __builtin_fma(p, r, c.c8);
  p = __builtin_fma(p, r, c.c7);
  p = __builtin_fma(p, r, c.c6);
  p = __builtin_fma(p, r, c.c5);
  p = __builtin_fma(p, r, c.c4);
  p = __builtin_fma(p, r, c.c3);
  p = __builtin_fma(p, r, c.c2);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_ldexp(p, (int)t);
}
constexpr double kExpBoundedRel = 0x1p-46;

AMWG_HD double lo_zeroed(double x) { return bits_f64(f64_bits(x) & 0xffffffff00000000ull); }
AMWG_HD double scalbn_v8(double x, int n) {
  const double two54 = 1.80143985094819840000e+16, twom54 = 5.55111512312578270212e-17, huge = 1.0e+300, tiny = 1.0e-300;
  int32_t hx = hi_word(x);
  const uint32_t lx = lo_word(x);
  int32_t k = (hx & 0x7ff00000) >> 20;
  if (k == 0) {
    if ((lx | (uint32_t)(hx & 0x7fffffff)) == 0) return x;
    x *= two54;
    hx = hi_word(x);
    k = ((hx & 0x7ff00000) >> 20) - 54;
    if (n < -50000) return tiny * x;
  }
  if (k == 0x7ff) return x + x;
  k = k + n;
  if (k > 0x7fe) return huge * (x < 0 ? -huge : huge);
  if (k > 0) return set_hi_word(x, (hx & (int32_t)0x800fffff) | (k << 20));
  if (k <= -54) return (n > 50000) ? huge * (x < 0 ? -huge : huge) : tiny * (x < 0 ? -tiny : tiny);
  k += 54;
  return set_hi_word(x, (hx & (int32_t)0x800fffff) | (k << 20)) * twom54;
}

AMWG_HD double pow_v8(double x, double y) {
  const double dp_h1 = 5.84962487220764160156e-01, dp_l1 = 1.35003920212974897128e-08;
  const double zero = 0.0, one = 1.0, two = 2.0, two53 = 9007199254740992.0, huge = 1.0e300, tiny = 1.0e-300,
    L1 = 5.99999999999994648725e-01, L2 = 4.28571428578550184252e-01, L3 = 3.33333329818377432918e-01,
    L4 = 2.72728123808534006489e-01, L5 = 2.30660745775561754067e-01, L6 = 2.06975017800338417784e-01,
    P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
    P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08,
    lg2 = 6.93147180559945286227e-01, lg2_h = 6.93147182464599609375e-01, lg2_l = -1.90465429995776804525e-09,
    ovt = 8.0085662595372944372e-0017, cp = 9.61796693925975554329e-01, cp_h = 9.61796700954437255859e-01,
    cp_l = -7.02846165095275826516e-09, ivln2 = 1.44269504088896338700e+00, ivln2_h = 1.44269502162933349609e+00,
    ivln2_l = 1.92596299112661746887e-08;
  double z, ax, z_h, z_l, p_h, p_l, y1, t1, t2, r, s, t, u, v, w;
  int32_t i, j, k, yisint, n;
  const int32_t hx = hi_word(x), hy = hi_word(y);
  const uint32_t lx = lo_word(x), ly = lo_word(y);
  int32_t ix = hx & 0x7fffffff;
  const int32_t iy = hy & 0x7fffffff;
  if ((iy | (int32_t)ly) == 0) return one;
  if (ix > 0x7ff00000 || ((ix == 0x7ff00000) && (lx != 0)) || iy > 0x7ff00000 || ((iy == 0x7ff00000) && (ly != 0))) return x + y;
  yisint = 0;
  if (hx < 0) {
    if (iy >= 0x43400000) yisint = 2;
    else if (iy >= 0x3ff00000) {
      k = (iy >> 20) - 0x3ff;
      if (k > 20) {
        j = (int32_t)(ly >> (52 - k));
        if ((uint32_t)((uint32_t)j << (52 - k)) == ly) yisint = 2 - (j & 1);
      } else if (ly == 0) {
        j = iy >> (20 - k);
        if ((j << (20 - k)) == iy) yisint = 2 - (j & 1);
      }
    }
  }
  if (ly == 0) {
    if (iy == 0x7ff00000) {
      if (((ix - 0x3ff00000) | (int32_t)lx) == 0) return y - y;   // (+-1) ** +-Infinity is NaN
      else if (ix >= 0x3ff00000) return (hy >= 0) ? y : zero;
      else return (hy < 0) ? -y : zero;
    }
    if (iy == 0x3ff00000) return (hy < 0) ? one / x : x;
    if (hy == 0x40000000) return x * x;
    if (hy == 0x3fe00000 && hx >= 0) return __builtin_sqrt(x);
  }
  ax = __builtin_fabs(x);
  if (lx == 0) {
    if (ix == 0x7ff00000 || ix == 0 || ix == 0x3ff00000) {
      z = ax;
      if (hy < 0) z = one / z;
      if (hx < 0) {
        if (((ix - 0x3ff00000) | yisint) == 0) z = (z - z) / (z - z);
        else if (yisint == 1) z = -z;
      }
      return z;
    }
  }
  n = (hx < 0) ? 0 : 1;
  if ((n | yisint) == 0) return (x - x) / (x - x);
  s = one;
  if ((n | (yisint - 1)) == 0) s = -one;
  if (iy > 0x41e00000) {
    if (iy > 0x43f00000) {
      if (ix <= 0x3fefffff) return (hy < 0) ? huge * huge : tiny * tiny;
      if (ix >= 0x3ff00000) return (hy > 0) ? huge * huge : tiny * tiny;
    }
    if (ix < 0x3fefffff) return (hy < 0) ? s * huge * huge : s * tiny * tiny;
    if (ix > 0x3ff00000) return (hy > 0) ? s * huge * huge : s * tiny * tiny;
    t = ax - one;
    w = (t * t) * (0.5 - t * (0.3333333333333333333333 - t * 0.25));
    u = ivln2_h * t;
    v = t * ivln2_l - w * ivln2;
    t1 = lo_zeroed(u + v);
    t2 = v - (t1 - u);
  } else {
    double ss, s2, s_h, s_l, t_h, t_l;
    n = 0;
    if (ix < 0x00100000) { ax *= two53; n -= 53; ix = hi_word(ax); }
    n += ((ix) >> 20) - 0x3ff;
    j = ix & 0x000fffff;
    ix = j | 0x3ff00000;
    if (j <= 0x3988E) k = 0;
    else if (j < 0xBB67A) k = 1;
    else { k = 0; n += 1; ix -= 0x00100000; }
    ax = set_hi_word(ax, ix);
    const double bpk = k ? 1.5 : 1.0;
    u = ax - bpk;
    v = one / (ax + bpk);
    ss = u * v;
    s_h = lo_zeroed(ss);
    t_h = bits_f64((uint64_t)(uint32_t)(((ix >> 1) | 0x20000000) + 0x00080000 + (k << 18)) << 32);
    t_l = ax - (t_h - bpk);
    s_l = v * ((u - s_h * t_h) - s_h * t_l);
    s2 = ss * ss;
    r = s2 * s2 * (L1 + s2 * (L2 + s2 * (L3 + s2 * (L4 + s2 * (L5 + s2 * L6)))));
    r += s_l * (s_h + ss);
    s2 = s_h * s_h;
    t_h = lo_zeroed(3.0 + s2 + r);
    t_l = r - ((t_h - 3.0) - s2);
    u = s_h * t_h;
    v = s_l * t_h + t_l * ss;
    p_h = lo_zeroed(u + v);
    p_l = v - (p_h - u);
    z_h = cp_h * p_h;
    z_l = cp_l * p_h + p_l * cp + (k ? dp_l1 : 0.0);
    t = (double)n;
    const double dph = k ? dp_h1 : 0.0;
    t1 = lo_zeroed(((z_h + z_l) + dph) + t);
    t2 = z_l - (((t1 - t) - dph) - z_h);
  }
  y1 = lo_zeroed(y);
  p_l = (y - y1) * t1 + y * t2;
  p_h = y1 * t1;
  z = p_l + p_h;
  j = hi_word(z);
  i = (int32_t)lo_word(z);
  if (j >= 0x40900000) {
    if (((j - 0x40900000) | i) != 0) return s * huge * huge;
    if (p_l + ovt > z - p_h) return s * huge * huge;
  } else if ((j & 0x7fffffff) >= 0x4090cc00) {
    if (((j - (int32_t)0xc090cc00) | i) != 0) return s * tiny * tiny;
    if (p_l <= z - p_h) return s * tiny * tiny;
  }
  i = j & 0x7fffffff;
  k = (i >> 20) - 0x3ff;
  n = 0;
  if (i > 0x3fe00000) {
    n = j + (0x00100000 >> (k + 1));
    k = ((n & 0x7fffffff) >> 20) - 0x3ff;
    t = bits_f64((uint64_t)(uint32_t)(n & ~(0x000fffff >> k)) << 32);
    n = ((n & 0x000fffff) | 0x00100000) >> (20 - k);
    if (j < 0) n = -n;
    p_h -= t;
  }
  t = lo_zeroed(p_l + p_h);
  u = t * lg2_h;
  v = (p_l - (t - p_h)) * lg2 + t * lg2_l;
  z = u + v;
  w = v - (z - u);
  t = z * z;
  t1 = z - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  r = (z * t1) / ((t1 - two) - (w + z * w));   // V8's grouping; fdlibm: (z*t1)/(t1-two) - (w+z*w)
  z = one - (r - z);
  j = hi_word(z);
  j += (n << 20);
  if ((j >> 20) <= 0) z = scalbn_v8(z, n);
  else z = set_hi_word(z, hi_word(z) + (n << 20));
  return s * z;
}

// ---- log1p(x), expm1(x): V8's Math.log1p / Math.expm1 (fdlibm s_log1p.c, s_expm1.c via src/base/ieee754.cc), for user closures
// (softplus, log-sum-exp).  Bit-identical to Node on 200 000 arguments (tests/golden/v8_log1p_expm1_pairs.bin).
AMWG_HD double log1p_v8(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, two54 = 1.80143985094819840000e+16,
    Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01, Lp4 = 2.222219843214978396e-01,
    Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01, Lp7 = 1.479819860511658591e-01, zero = 0.0;
  double hfsq, f = 0, c = 0, s, z, R, u;
  int32_t k, hx, hu = 0, ax;
  hx = hi_word(x); ax = hx & 0x7fffffff;
  k = 1;
  if (hx < 0x3FDA827A) {
    if (ax >= 0x3ff00000) { if (x == -1.0) return -two54 / zero; else return (x - x) / (x - x); }
    if (ax < 0x3e200000) { if (two54 + x > zero && ax < 0x3c900000) return x; else return x - x * x * 0.5; }
    if (hx > 0 || hx <= ((int32_t)0xbfd2bec3)) { k = 0; f = x; hu = 1; }
  }
  if (hx >= 0x7ff00000) return x + x;
  if (k != 0) {
    if (hx < 0x43400000) { u = 1.0 + x; hu = hi_word(u); k = (hu >> 20) - 1023; c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0); c /= u; }
    else { u = x; hu = hi_word(u); k = (hu >> 20) - 1023; c = 0; }
    hu &= 0x000fffff;
    if (hu < 0x6a09e) { u = set_hi_word(u, hu | 0x3ff00000); }
    else { k += 1; u = set_hi_word(u, hu | 0x3fe00000); hu = (0x00100000 - hu) >> 2; }
    f = u - 1.0;
  }
  hfsq = 0.5 * f * f;
  if (hu == 0) {
    if (f == zero) { if (k == 0) return zero; else { c += k * ln2_lo; return k * ln2_hi + c; } }
    R = hfsq * (1.0 - 0.66666666666666666 * f);
    if (k == 0) return f - R; else return k * ln2_hi - ((R - (k * ln2_lo + c)) - f);
  }
  s = f / (2.0 + f);
  z = s * s;
  R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
  if (k == 0) return f - (hfsq - s * (hfsq + R)); else return k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + c))) - f);
}
// softplus(x) = log1p_v8(exp_v8(x)) -- the log-likelihood of a logistic regression written `y*eta - Math.log1p(Math.exp(eta))` -- as ONE
// straight line for a 64-lane SIMD: fdlibm's log1p has four data-dependent branches (x < sqrt(2)-1: no reduction; the correction term c
// of u = 1 + x by the exponent of u; which half of [sqrt(2)/2, sqrt(2)) the significand lands in; |f| < 2^-20) and two IEEE divisions, and a
// wavefront whose lanes are different observations takes every side of every branch.  Here the branches are selects, both quotients
// are quot_plain (their operands need no exponent juggling: 2 + f in (1.7, 2.42); u in [1.41, 2^53) and c zero or >= 2^-54 in
// magnitude), and what is left -- exp's rare arguments, exp(x) < 2^-29 (log1p's small-argument forms) or >= 2^53 (its large-argument form),
// |f| < 2^-20 -- leaves through one rarely taken branch to the full functions.  For exp(x) >= sqrt(2)-1 the exponent k of the reduced
// argument is >= 1 (u >= sqrt(2): its significand's high word is >= 0x6a09e, so k is raised), which is why only TWO result forms remain:
// k == 0 exactly when no reduction was made.  Same bits as log1p_v8(exp_v8(x)) (tests/host/explog_fuzz.cpp, tests/test_gpu_math.py).
#if defined(AMWG_X_NOCOLD)
AMWG_HD double log1p_exp_cold(double x) { return log1p_v8(exp_v8(x)); }
#else
AMWG_HD_OUTLINE double log1p_exp_cold(double x) { return log1p_v8(exp_v8(x)); }
#endif
template <class K>
AMWG_HD double log1p_exp_v8(double x, const K &c) {
  // exp(-20) = 2.06e-9 > 2^-29 = 1.86e-9 and exp(36) = 4.3e15 < 2^53 = 9.0e15 with room to spare for exp's last-place error; a NaN fails the first test
  if (__builtin_expect(!(x >= -20.0 && x <= 36.0) || exp_is_rare(x), 0)) return log1p_exp_cold(x);
  const ExpParts e = exp_parts(x, c);
  const double v = set_hi_word(e.y, hi_word(e.y) + (e.k << 20));      // exp(x), in (2^-29, 2^53)
  const bool small = hi_word(v) < 0x3FDA827A;                            // below sqrt(2) - 1: f = v, k = 0
  const double u = 1.0 + v;
  const int32_t hu0 = hi_word(u);
  const int32_t ke = (hu0 >> 20) - 1023;                                 // >= 0 (u > 1)
  const double cn = (ke > 0) ? 1.0 - (u - v) : v - (u - 1.0);            // the rounding error of u = 1 + v
  const double cq = quot_plain(cn, u);
  const int32_t mant = hu0 & 0x000fffff;
  const bool up = mant >= 0x6a09e;
  const double un = set_hi_word(u, mant | (up ? 0x3fe00000 : 0x3ff00000));
  // |f| < 2^-20 after the reduction (fdlibm's `hu == 0`: significand high word 0, or within 3 of the next power of two)
  if (__builtin_expect(!small && (up ? mant > 0xffffc : mant == 0), 0)) return log1p_exp_cold(x);
  const double f = small ? v : un - 1.0;
  const double dk = (double)(ke + (up ? 1 : 0));
  const double hfsq = 0.5 * f * f;
  const double s = quot_plain(f, 2.0 + f);
  const double z = s * s;
  const double R = z * (c.Lg1 + z * (c.Lg2 + z * (c.Lg3 + z * (c.Lg4 + z * (c.Lg5 + z * (c.Lg6 + z * c.Lg7))))));   // (log1p's Lp1..Lp7 are log's Lg1..Lg7)
  const double sr = s * (hfsq + R);
  const double r0 = f - (hfsq - sr);
  const double rk = dk * c.ln2_hi - ((hfsq - (sr + (dk * c.ln2_lo + cq))) - f);
  return small ? r0 : rk;
}
AMWG_HD double log1p_exp_v8(double x) { return log1p_exp_v8(x, ExpLogLiterals{}); }
// The same without its two branches, for an unrolled loop: `rare` is SET (never cleared) where the argument needs the full functions and the
// value returned is then garbage (no trap); the caller evaluates its U terms back to back -- U independent dependency chains in one basic
// block --, tests `rare` ONCE and re-forms the terms of such a lane with log1p_exp_cold (translate.js emits exactly that).
AMWG_HD double log1p_exp_v8_open(bool &rare, double x) {
  const ExpLogLiterals c{};
  const ExpParts e = exp_parts(x, c);
  const double v = set_hi_word(e.y, hi_word(e.y) + (e.k << 20));
  const bool small = hi_word(v) < 0x3FDA827A;
  const double u = 1.0 + v;
  const int32_t hu0 = hi_word(u);
  const int32_t ke = (hu0 >> 20) - 1023;
  const double cn = (ke > 0) ? 1.0 - (u - v) : v - (u - 1.0);
  const double cq = quot_plain(cn, u);
  const int32_t mant = hu0 & 0x000fffff;
  const bool up = mant >= 0x6a09e;
  const double un = set_hi_word(u, mant | (up ? 0x3fe00000 : 0x3ff00000));
  rare = rare || !(x >= -20.0 && x <= 36.0) || exp_is_rare(x) || (!small && (up ? mant > 0xffffc : mant == 0));
  const double f = small ? v : un - 1.0;
  const double dk = (double)(ke + (up ? 1 : 0));
  const double hfsq = 0.5 * f * f;
  const double s = quot_plain(f, 2.0 + f);
  const double z = s * s;
  const double R = z * (c.Lg1 + z * (c.Lg2 + z * (c.Lg3 + z * (c.Lg4 + z * (c.Lg5 + z * (c.Lg6 + z * c.Lg7))))));
  const double sr = s * (hfsq + R);
  const double r0 = f - (hfsq - sr);
  const double rk = dk * c.ln2_hi - ((hfsq - (sr + (dk * c.ln2_lo + cq))) - f);
  return small ? r0 : rk;
}

AMWG_HD double expm1_v8(double x) {
  const double one = 1.0, huge = 1.0e+300, tiny = 1.0e-300, o_threshold = 7.09782712893383973096e+02,
    ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00,
    Q1 = -3.33333333333331316428e-02, Q2 = 1.58730158725481460165e-03, Q3 = -7.93650757867487942473e-05,
    Q4 = 4.00821782732936239552e-06, Q5 = -2.01099218183624371326e-07;
  double y, hi, lo, c = 0, t, e, hxs, hfx, r1;
  int32_t k, xsb;
  uint32_t hx;
  hx = (uint32_t)hi_word(x);
  xsb = hx & 0x80000000;
  hx &= 0x7fffffff;
  if (hx >= 0x4043687A) {
    if (hx >= 0x40862E42) {
      if (hx >= 0x7ff00000) { if (((hx & 0xfffff) | lo_word(x)) != 0) return x + x; else return (xsb == 0) ? x : -1.0; }
      if (x > o_threshold) return huge * huge;
    }
    if (xsb != 0) { if (x + tiny < 0.0) return tiny - one; }
  }
  if (hx > 0x3fd62e42) {
    if (hx < 0x3FF0A2B2) {
      if (xsb == 0) { hi = x - ln2_hi; lo = ln2_lo; k = 1; } else { hi = x + ln2_hi; lo = -ln2_lo; k = -1; }
    } else {
      k = (int32_t)(invln2 * x + ((xsb == 0) ? 0.5 : -0.5));
      t = k; hi = x - t * ln2_hi; lo = t * ln2_lo;
    }
    x = hi - lo;
    c = (hi - x) - lo;
  } else if (hx < 0x3c900000) {
    t = huge + x;
    return x - (t - (huge + x));
  } else k = 0;
  hfx = 0.5 * x;
  hxs = x * hfx;
  r1 = one + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
  t = 3.0 - r1 * hfx;
  e = hxs * ((r1 - t) / (6.0 - x * t));
  if (k == 0) return x - (x * e - hxs);
  e = (x * (e - c) - c);
  e -= hxs;
  if (k == -1) return 0.5 * (x - e) - 0.5;
  if (k == 1) { if (x < -0.25) return -2.0 * (e - (x + 0.5)); else return one + 2.0 * (x - e); }
  if (k <= -2 || k > 56) {
    y = one - (e - x);
    if (k == 1024) y = y * 2.0 * 8.98846567431157953865e+307; else y = set_hi_word(y, hi_word(y) + (k << 20));
    return y - one;
  }
  t = one;
  if (k < 20) { t = set_hi_word(t, 0x3ff00000 - (0x200000 >> k)); y = t - (e - x); y = set_hi_word(y, hi_word(y) + (k << 20)); }
  else { t = set_hi_word(t, ((0x3ff - k) << 20)); y = x - (e + t); y += one; y = set_hi_word(y, hi_word(y) + (k << 20)); }
  return y;
}

// ---- tanh, atan, log10: V8's Math.tanh / Math.atan / Math.log10 (fdlibm s_tanh.c, s_atan.c, e_log10.c), for user closures.
// Bit-identical to Node on 200 000 arguments each (tests/golden/v8_math2_pairs.bin).
AMWG_HD double tanh_v8(double x) {
  const double one = 1.0, two = 2.0, tiny = 1.0e-300, huge = 1.0e300;
  double t, z;
  int32_t jx = hi_word(x), ix = jx & 0x7fffffff;
  if (ix >= 0x7ff00000) { if (jx >= 0) return one / x + one; else return one / x - one; }
  if (ix < 0x40360000) {            /* |x| < 22 */
    if (ix < 0x3e300000) { if (huge + x > one) return x; }   /* |x| < 2**-28 */
    if (ix >= 0x3ff00000) { t = expm1_v8(two * __builtin_fabs(x)); z = one - two / (t + two); }
    else { t = expm1_v8(-two * __builtin_fabs(x)); z = -t / (t + two); }
  } else z = one - tiny;
  return (jx >= 0) ? z : -z;
}
AMWG_HD double atan_v8(double x) {
  const double atanhi[] = {4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01, 1.57079632679489655800e+00};
  const double atanlo[] = {2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17, 6.12323399573676603587e-17};
  const double aT[] = {3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01, -1.11111104054623557880e-01,
    9.09088713343650656196e-02, -7.69187620504482999495e-02, 6.66107313738753120669e-02, -5.83357013379057348645e-02,
    4.97687799461593236017e-02, -3.65315727442169155270e-02, 1.62858201153657823623e-02};
  const double one = 1.0, huge = 1.0e300;
  double w, s1, s2, z;
  int32_t ix, hx, id;
  hx = hi_word(x); ix = hx & 0x7fffffff;
  if (ix >= 0x44100000) {
    if (ix > 0x7ff00000 || (ix == 0x7ff00000 && (lo_word(x) != 0))) return x + x;
    if (hx > 0) return atanhi[3] + atanlo[3]; else return -atanhi[3] - atanlo[3];
  }
  if (ix < 0x3fdc0000) {
    if (ix < 0x3e200000) { if (huge + x > one) return x; }
    id = -1;
  } else {
    x = __builtin_fabs(x);
    if (ix < 0x3ff30000) {
      if (ix < 0x3fe60000) { id = 0; x = (2.0 * x - one) / (2.0 + x); }
      else { id = 1; x = (x - one) / (x + one); }
    } else {
      if (ix < 0x40038000) { id = 2; x = (x - 1.5) / (one + 1.5 * x); }
      else { id = 3; x = -1.0 / x; }
    }
  }
  z = x * x;
  w = z * z;
  s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  if (id < 0) return x - x * (s1 + s2);
  z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
  return (hx < 0) ? -z : z;
}
AMWG_HD double log10_v8(double x) {
  const double two54 = 1.80143985094819840000e+16, ivln10 = 4.34294481903251816668e-01, log10_2hi = 3.01029995663611771306e-01, log10_2lo = 3.69423907715893078616e-13, zero = 0.0;
  double y, z;
  int32_t i, k, hx;
  uint32_t lx;
  hx = hi_word(x); lx = lo_word(x);
  k = 0;
  if (hx < 0x00100000) {
    if (((hx & 0x7fffffff) | lx) == 0) return -two54 / zero;
    if (hx < 0) return (x - x) / zero;
    k -= 54; x *= two54; hx = hi_word(x);
  }
  if (hx >= 0x7ff00000) return x + x;
  k += (hx >> 20) - 1023;
  i = ((uint32_t)k & 0x80000000) >> 31;
  hx = (hx & 0x000fffff) | ((0x3ff - i) << 20);
  y = (double)(k + i);
  x = set_hi_word(x, hx);
  z = y * log10_2lo + ivln10 * log_v8(x);
  return z + y * log10_2hi;
}

// Math.round: nearest integer, ties toward +infinity (mcmc.js:597); a result of zero keeps the sign of x (ECMA-262: -0 for
// -0.5 <= x <= -0), which the state then carries: an int parameter proposed in [-0.5, 0) sits at -0, as in the reference.
AMWG_HD double js_round(double x) {
  if (!(__builtin_fabs(x) < 4503599627370496.0)) return x;
  const double f = __builtin_floor(x);
  return __builtin_copysign((x - f >= 0.5) ? f + 1.0 : f, x);
}

}  // namespace amwg

#include "amwg_trig.h"
