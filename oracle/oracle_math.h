/*
 * oracle_math.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * fp64 exp() and log() that reproduce, bit for bit, what the reference's
 * JavaScript engine returns for Math.exp / Math.log.  The reference
 * (rasmusab/bayes.js) has no arithmetic of its own beyond V8's Math.*:
 *   - mcmc.js:51   Math.log  (rnorm rejection test)
 *   - mcmc.js:527  Math.exp  (accept probability)
 *   - mcmc.js:578  Math.exp  (proposal sd = exp(prop_log_scale))
 *   - distributions.js:94-99 log/exp aliases used by every ld.* density
 * V8 (third-party, absent from /root/reference; Node v12.22.9 => V8 7.x,
 * src/base/ieee754.cc) implements both with the Sun fdlibm algorithms
 * (e_exp.c, e_log.c, "Developed at SunSoft ... Permission to use, copy, modify,
 * and distribute this software is freely granted").  The published algorithms are
 * restated here; tests/test_oracle_math.py pins them against 2e6 outputs of
 * Node's own Math.exp/Math.log (tests/golden/v8_math_pairs.bin, generated
 * by oracle/gen_math_pairs.js).
 *
 * Compile with -ffp-contract=off: fdlibm's error analysis assumes every
 * operation rounds once.
 */
#ifndef AMWG_ORACLE_MATH_H
#define AMWG_ORACLE_MATH_H

#include <stdint.h>
#include <string.h>

static inline uint64_t om_bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double om_from_bits(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
static inline int32_t om_hi(double x) { return (int32_t)(om_bits(x) >> 32); }
static inline uint32_t om_lo(double x) { return (uint32_t)om_bits(x); }
static inline double om_with_hi(double x, int32_t hi) {
  return om_from_bits(((uint64_t)(uint32_t)hi << 32) | (uint64_t)om_lo(x));
}

/* exp(x): x = k*ln2 + r, |r| <= 0.5 ln2; exp(r) from the degree-5 Remez
 * polynomial for R(r^2) = r*(exp(r)+1)/(exp(r)-1); scale by 2^k. */
static double om_exp(double x) {
  static const double half_pm[2] = {0.5, -0.5};
  static const double ln2_hi[2] = {6.93147180369123816490e-01, -6.93147180369123816490e-01};
  static const double ln2_lo[2] = {1.90821492927058770002e-10, -1.90821492927058770002e-10};
  const double huge = 1.0e+300;
  const double two_m1000 = 9.33263618503218878990e-302;
  const double overflow_at = 7.09782712893383973096e+02;
  const double underflow_at = -7.45133219101941108420e+02;
  const double inv_ln2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
               P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
               P5 = 4.13813679705723846039e-08;
  double hi = 0.0, lo = 0.0, c, t, y;
  int32_t k = 0;
  uint32_t hx = (uint32_t)om_hi(x);
  int sign = (int)(hx >> 31);
  hx &= 0x7fffffffu;

  if (hx >= 0x40862E42u) {            /* |x| >= 709.78 or non-finite */
    if (hx >= 0x7ff00000u) {
      if (((hx & 0xfffffu) | om_lo(x)) != 0) return x + x;   /* NaN */
      return sign ? 0.0 : x;                                 /* exp(+-inf) */
    }
    if (x > overflow_at) return huge * huge;
    if (x < underflow_at) return two_m1000 * two_m1000;
  }
  if (hx > 0x3fd62e42u) {             /* |x| > 0.5 ln2 */
    if (hx < 0x3FF0A2B2u) {           /* and |x| < 1.5 ln2 */
      /* V8 special-cases exp(1) to return Math.E exactly (the polynomial is 1 ulp high there);
       * found by the x = 1 record of tests/golden/v8_math_pairs.bin */
      if (x == 1.0) return 2.718281828459045;
      hi = x - ln2_hi[sign];
      lo = ln2_lo[sign];
      k = 1 - sign - sign;
    } else {
      k = (int32_t)(inv_ln2 * x + half_pm[sign]);
      t = (double)k;
      hi = x - t * ln2_hi[0];
      lo = t * ln2_lo[0];
    }
    x = hi - lo;
  } else if (hx < 0x3e300000u) {      /* |x| < 2^-28 */
    if (huge + x > 1.0) return 1.0 + x;
  } else {
    k = 0;
  }
  t = x * x;
  c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
  y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
  if (k >= -1021) {
    if (k == 1024) return y * 2.0 * 8.98846567431157953865e+307; /* 2^1023 */
    return om_with_hi(y, om_hi(y) + (k << 20));
  }
  y = om_with_hi(y, om_hi(y) + ((k + 1000) << 20));
  return y * two_m1000;
}

/* log(x): x = 2^k (1+f), sqrt(2)/2 < 1+f < sqrt(2); s = f/(2+f);
 * log(1+f) = f - s*(f - R(s^2)) with a degree-14 polynomial R. */
static double om_log(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double two54 = 1.80143985094819840000e+16;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
               Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
               Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  const double zero = 0.0;
  double hfsq, f, s, z, R, w, t1, t2, dk;
  int32_t k = 0, hx = om_hi(x), i, j;
  uint32_t lx = om_lo(x);

  if (hx < 0x00100000) {              /* x < 2^-1022: zero, negative or subnormal */
    if (((hx & 0x7fffffff) | (int32_t)lx) == 0) return -two54 / zero;
    if (hx < 0) return (x - x) / zero;
    k -= 54;
    x *= two54;
    hx = om_hi(x);
  }
  if (hx >= 0x7ff00000) return x + x;
  k += (hx >> 20) - 1023;
  hx &= 0x000fffff;
  i = (hx + 0x95f64) & 0x100000;
  x = om_with_hi(x, hx | (i ^ 0x3ff00000));   /* x or x/2 in [sqrt(2)/2, sqrt(2)) */
  k += (i >> 20);
  f = x - 1.0;
  if ((0x000fffff & (2 + hx)) < 3) {  /* |f| < 2^-20 */
    if (f == zero) {
      if (k == 0) return zero;
      dk = (double)k;
      return dk * ln2_hi + dk * ln2_lo;
    }
    R = f * f * (0.5 - 0.33333333333333333 * f);
    if (k == 0) return f - R;
    dk = (double)k;
    return dk * ln2_hi - ((R - dk * ln2_lo) - f);
  }
  s = f / (2.0 + f);
  dk = (double)k;
  z = s * s;
  i = hx - 0x6147a;
  w = z * z;
  j = 0x6b851 - hx;
  t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  i |= j;
  R = t2 + t1;
  if (i > 0) {
    hfsq = 0.5 * f * f;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  }
  if (k == 0) return f - s * (f - R);
  return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

#endif
