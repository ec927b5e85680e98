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

#include <math.h>
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


/* pow(x, y): fdlibm e_pow.c as carried by V8 (src/base/ieee754.cc, ieee754::pow), which is what
 * Math.pow evaluates on the reference's engine (distributions.js:98 `pow = Math.pow`; used with
 * general exponents by ld.t :185-189 and ld.weibull :194-200).  log2(x) in two pieces
 * (t1 + t2), y*log2(x) in two pieces, then 2^(p_h + p_l).  ECMAScript deviates from C in one
 * place: (+-1) ** (+-Infinity) is NaN (fdlibm's `y - y`), and V8's port differs from fdlibm in one
 * grouping (marked below).  Pinned against Node's own Math.pow by
 * tests/test_oracle_math.py (tests/golden/v8_pow_pairs.bin, oracle/gen_math_pairs.js). */
static double om_scalbn(double x, int n) { /* fdlibm s_scalbn.c */
  static const double two54 = 1.80143985094819840000e+16, twom54 = 5.55111512312578270212e-17, huge = 1.0e+300, tiny = 1.0e-300;
  int32_t hx = om_hi(x), k;
  uint32_t lx = om_lo(x);
  k = (hx & 0x7ff00000) >> 20;
  if (k == 0) {
    if ((lx | (uint32_t)(hx & 0x7fffffff)) == 0) return x;
    x *= two54;
    hx = om_hi(x);
    k = ((hx & 0x7ff00000) >> 20) - 54;
    if (n < -50000) return tiny * x;
  }
  if (k == 0x7ff) return x + x;
  k = k + n;
  if (k > 0x7fe) return huge * (x < 0 ? -huge : huge);
  if (k > 0) return om_with_hi(x, (hx & (int32_t)0x800fffff) | (k << 20));
  if (k <= -54) {
    if (n > 50000) return huge * (x < 0 ? -huge : huge);
    return tiny * (x < 0 ? -tiny : tiny);
  }
  k += 54;
  x = om_with_hi(x, (hx & (int32_t)0x800fffff) | (k << 20));
  return x * twom54;
}

static double om_sqrt(double x) { return sqrt(x); }   /* IEEE, correctly rounded */
static double om_pow(double x, double y) {
  static const double bp[2] = {1.0, 1.5}, dp_h[2] = {0.0, 5.84962487220764160156e-01}, dp_l[2] = {0.0, 1.35003920212974897128e-08};
  static const double zero = 0.0, one = 1.0, two = 2.0, two53 = 9007199254740992.0, huge = 1.0e300, tiny = 1.0e-300,
    L1 = 5.99999999999994648725e-01, L2 = 4.28571428578550184252e-01, L3 = 3.33333329818377432918e-01,
    L4 = 2.72728123808534006489e-01, L5 = 2.30660745775561754067e-01, L6 = 2.06975017800338417784e-01,
    P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
    P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08,
    lg2 = 6.93147180559945286227e-01, lg2_h = 6.93147182464599609375e-01, lg2_l = -1.90465429995776804525e-09,
    ovt = 8.0085662595372944372e-0017, cp = 9.61796693925975554329e-01, cp_h = 9.61796700954437255859e-01,
    cp_l = -7.02846165095275826516e-09, ivln2 = 1.44269504088896338700e+00, ivln2_h = 1.44269502162933349609e+00,
    ivln2_l = 1.92596299112661746887e-08;
  double z, ax, z_h, z_l, p_h, p_l, y1, t1, t2, r, s, t, u, v, w;
  int32_t i, j, k, yisint, n, hx, hy, ix, iy;
  uint32_t lx, ly;
  hx = om_hi(x); lx = om_lo(x);
  hy = om_hi(y); ly = om_lo(y);
  ix = hx & 0x7fffffff; iy = hy & 0x7fffffff;
  if ((iy | ly) == 0) return one;
  if (ix > 0x7ff00000 || ((ix == 0x7ff00000) && (lx != 0)) || iy > 0x7ff00000 || ((iy == 0x7ff00000) && (ly != 0))) return x + y;
  yisint = 0;
  if (hx < 0) {
    if (iy >= 0x43400000) yisint = 2;
    else if (iy >= 0x3ff00000) {
      k = (iy >> 20) - 0x3ff;
      if (k > 20) {
        j = (int32_t)(ly >> (52 - k));
        if ((uint32_t)(j << (52 - k)) == ly) yisint = 2 - (j & 1);
      } else if (ly == 0) {
        j = iy >> (20 - k);
        if ((j << (20 - k)) == iy) yisint = 2 - (j & 1);
      }
    }
  }
  if (ly == 0) {
    if (iy == 0x7ff00000) {
      if (((ix - 0x3ff00000) | lx) == 0) return y - y;          /* (+-1)**+-inf is NaN (also the ECMAScript rule) */
      else if (ix >= 0x3ff00000) return (hy >= 0) ? y : zero;
      else return (hy < 0) ? -y : zero;
    }
    if (iy == 0x3ff00000) { if (hy < 0) return one / x; else return x; }
    if (hy == 0x40000000) return x * x;
    if (hy == 0x3fe00000) { if (hx >= 0) return om_sqrt(x); }
  }
  ax = x < 0 ? -x : x;
  if (om_hi(ax) < 0) ax = -ax; /* -0 */
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
  n = (hx < 0) ? 0 : 1;   /* fdlibm: (hx >> 31) + 1 */
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
    t1 = u + v;
    t1 = om_from_bits(om_bits(t1) & 0xffffffff00000000ull);
    t2 = v - (t1 - u);
  } else {
    double ss, s2, s_h, s_l, t_h, t_l;
    n = 0;
    if (ix < 0x00100000) { ax *= two53; n -= 53; ix = om_hi(ax); }
    n += ((ix) >> 20) - 0x3ff;
    j = ix & 0x000fffff;
    ix = j | 0x3ff00000;
    if (j <= 0x3988E) k = 0;
    else if (j < 0xBB67A) k = 1;
    else { k = 0; n += 1; ix -= 0x00100000; }
    ax = om_with_hi(ax, ix);
    u = ax - bp[k];
    v = one / (ax + bp[k]);
    ss = u * v;
    s_h = om_from_bits(om_bits(ss) & 0xffffffff00000000ull);
    t_h = om_from_bits((uint64_t)(uint32_t)(((ix >> 1) | 0x20000000) + 0x00080000 + (k << 18)) << 32);
    t_l = ax - (t_h - bp[k]);
    s_l = v * ((u - s_h * t_h) - s_h * t_l);
    s2 = ss * ss;
    r = s2 * s2 * (L1 + s2 * (L2 + s2 * (L3 + s2 * (L4 + s2 * (L5 + s2 * L6)))));
    r += s_l * (s_h + ss);
    s2 = s_h * s_h;
    t_h = 3.0 + s2 + r;
    t_h = om_from_bits(om_bits(t_h) & 0xffffffff00000000ull);
    t_l = r - ((t_h - 3.0) - s2);
    u = s_h * t_h;
    v = s_l * t_h + t_l * ss;
    p_h = u + v;
    p_h = om_from_bits(om_bits(p_h) & 0xffffffff00000000ull);
    p_l = v - (p_h - u);
    z_h = cp_h * p_h;
    z_l = cp_l * p_h + p_l * cp + dp_l[k];
    t = (double)n;
    t1 = (((z_h + z_l) + dp_h[k]) + t);
    t1 = om_from_bits(om_bits(t1) & 0xffffffff00000000ull);
    t2 = z_l - (((t1 - t) - dp_h[k]) - z_h);
  }
  y1 = om_from_bits(om_bits(y) & 0xffffffff00000000ull);
  p_l = (y - y1) * t1 + y * t2;
  p_h = y1 * t1;
  z = p_l + p_h;
  j = om_hi(z);
  i = (int32_t)om_lo(z);
  if (j >= 0x40900000) {
    if (((j - 0x40900000) | i) != 0) return s * huge * huge;
    else { if (p_l + ovt > z - p_h) return s * huge * huge; }
  } else if ((j & 0x7fffffff) >= 0x4090cc00) {
    if (((j - (int32_t)0xc090cc00) | i) != 0) return s * tiny * tiny;
    else { if (p_l <= z - p_h) return s * tiny * tiny; }
  }
  i = j & 0x7fffffff;
  k = (i >> 20) - 0x3ff;
  n = 0;
  if (i > 0x3fe00000) {
    n = j + (0x00100000 >> (k + 1));
    k = ((n & 0x7fffffff) >> 20) - 0x3ff;
    t = om_from_bits((uint64_t)(uint32_t)(n & ~(0x000fffff >> k)) << 32);
    n = ((n & 0x000fffff) | 0x00100000) >> (20 - k);
    if (j < 0) n = -n;
    p_h -= t;
  }
  t = p_l + p_h;
  t = om_from_bits(om_bits(t) & 0xffffffff00000000ull);
  u = t * lg2_h;
  v = (p_l - (t - p_h)) * lg2 + t * lg2_l;
  z = u + v;
  w = v - (z - u);
  t = z * z;
  t1 = z - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  /* V8's port groups this as z*t1 / ((t1-2) - (w+z*w)) where fdlibm has (z*t1)/(t1-2) - (w+z*w);
   * Node's Math.pow follows V8 (300 000 pairs, tests/golden/v8_pow_pairs.bin), so does this. */
  r = (z * t1) / ((t1 - two) - (w + z * w));
  z = one - (r - z);
  j = om_hi(z);
  j += (n << 20);
  if ((j >> 20) <= 0) z = om_scalbn(z, n);
  else z = om_with_hi(z, om_hi(z) + (n << 20));
  return s * z;
}

/* log1p(x), expm1(x): fdlibm s_log1p.c / s_expm1.c as carried by V8 (src/base/ieee754.cc), i.e. Math.log1p / Math.expm1 of the
 * reference's engine (user closures may call them; distributions.js itself does not).  Pinned against Node's own outputs by
 * tests/test_oracle_math.py (tests/golden/v8_log1p_expm1_pairs.bin, oracle/gen_ld_golden.js). */
static double om_log1p(double x) {
  static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, two54 = 1.80143985094819840000e+16,
    Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01, Lp4 = 2.222219843214978396e-01,
    Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01, Lp7 = 1.479819860511658591e-01, zero = 0.0;
  double hfsq, f = 0, c = 0, s, z, R, u;
  int32_t k, hx, hu = 0, ax;
  hx = om_hi(x); ax = hx & 0x7fffffff;
  k = 1;
  if (hx < 0x3FDA827A) {
    if (ax >= 0x3ff00000) { if (x == -1.0) return -two54 / zero; else return (x - x) / (x - x); }
    if (ax < 0x3e200000) { if (two54 + x > zero && ax < 0x3c900000) return x; else return x - x * x * 0.5; }
    if (hx > 0 || hx <= ((int32_t)0xbfd2bec3)) { k = 0; f = x; hu = 1; }
  }
  if (hx >= 0x7ff00000) return x + x;
  if (k != 0) {
    if (hx < 0x43400000) { u = 1.0 + x; hu = om_hi(u); k = (hu >> 20) - 1023; c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0); c /= u; }
    else { u = x; hu = om_hi(u); k = (hu >> 20) - 1023; c = 0; }
    hu &= 0x000fffff;
    if (hu < 0x6a09e) { u = om_with_hi(u, hu | 0x3ff00000); }
    else { k += 1; u = om_with_hi(u, hu | 0x3fe00000); hu = (0x00100000 - hu) >> 2; }
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
static double om_expm1(double x) {
  static const double one = 1.0, huge = 1.0e+300, tiny = 1.0e-300, o_threshold = 7.09782712893383973096e+02,
    ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00,
    Q1 = -3.33333333333331316428e-02, Q2 = 1.58730158725481460165e-03, Q3 = -7.93650757867487942473e-05,
    Q4 = 4.00821782732936239552e-06, Q5 = -2.01099218183624371326e-07;
  double y, hi, lo, c = 0, t, e, hxs, hfx, r1;
  int32_t k, xsb;
  uint32_t hx;
  hx = (uint32_t)om_hi(x);
  xsb = hx & 0x80000000;
  hx &= 0x7fffffff;
  if (hx >= 0x4043687A) {
    if (hx >= 0x40862E42) {
      if (hx >= 0x7ff00000) { if (((hx & 0xfffff) | om_lo(x)) != 0) return x + x; else return (xsb == 0) ? x : -1.0; }
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
    if (k == 1024) y = y * 2.0 * 8.98846567431157953865e+307; else y = om_with_hi(y, om_hi(y) + (k << 20));
    return y - one;
  }
  t = one;
  if (k < 20) { t = om_with_hi(t, 0x3ff00000 - (0x200000 >> k)); y = t - (e - x); y = om_with_hi(y, om_hi(y) + (k << 20)); }
  else { t = om_with_hi(t, ((0x3ff - k) << 20)); y = x - (e + t); y += one; y = om_with_hi(y, om_hi(y) + (k << 20)); }
  return y;
}

/* tanh(x), atan(x), log10(x): fdlibm s_tanh.c (on expm1), s_atan.c, e_log10.c as carried by V8 -- Math.tanh / Math.atan /
 * Math.log10 for user closures.  Pinned against Node's outputs (tests/golden/v8_math2_pairs.bin, 60 000 arguments). */
static double om_tanh(double x) {
  static const double one = 1.0, two = 2.0, tiny = 1.0e-300, huge = 1.0e300;
  double t, z;
  int32_t jx = om_hi(x), ix = jx & 0x7fffffff;
  if (ix >= 0x7ff00000) { if (jx >= 0) return one / x + one; else return one / x - one; }
  if (ix < 0x40360000) {            /* |x| < 22 */
    if (ix < 0x3e300000) { if (huge + x > one) return x; }   /* |x| < 2**-28 */
    if (ix >= 0x3ff00000) { t = om_expm1(two * fabs(x)); z = one - two / (t + two); }
    else { t = om_expm1(-two * fabs(x)); z = -t / (t + two); }
  } else z = one - tiny;
  return (jx >= 0) ? z : -z;
}
static double om_atan(double x) {
  static const double atanhi[] = {4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01, 1.57079632679489655800e+00};
  static const double atanlo[] = {2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17, 6.12323399573676603587e-17};
  static const double aT[] = {3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01, -1.11111104054623557880e-01,
    9.09088713343650656196e-02, -7.69187620504482999495e-02, 6.66107313738753120669e-02, -5.83357013379057348645e-02,
    4.97687799461593236017e-02, -3.65315727442169155270e-02, 1.62858201153657823623e-02};
  static const double one = 1.0, huge = 1.0e300;
  double w, s1, s2, z;
  int32_t ix, hx, id;
  hx = om_hi(x); ix = hx & 0x7fffffff;
  if (ix >= 0x44100000) {
    if (ix > 0x7ff00000 || (ix == 0x7ff00000 && (om_lo(x) != 0))) return x + x;
    if (hx > 0) return atanhi[3] + atanlo[3]; else return -atanhi[3] - atanlo[3];
  }
  if (ix < 0x3fdc0000) {
    if (ix < 0x3e200000) { if (huge + x > one) return x; }
    id = -1;
  } else {
    x = fabs(x);
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
static double om_log10(double x) {
  static const double two54 = 1.80143985094819840000e+16, ivln10 = 4.34294481903251816668e-01, log10_2hi = 3.01029995663611771306e-01, log10_2lo = 3.69423907715893078616e-13, zero = 0.0;
  double y, z;
  int32_t i, k, hx;
  uint32_t lx;
  hx = om_hi(x); lx = om_lo(x);
  k = 0;
  if (hx < 0x00100000) {
    if (((hx & 0x7fffffff) | lx) == 0) return -two54 / zero;
    if (hx < 0) return (x - x) / zero;
    k -= 54; x *= two54; hx = om_hi(x);
  }
  if (hx >= 0x7ff00000) return x + x;
  k += (hx >> 20) - 1023;
  i = ((uint32_t)k & 0x80000000) >> 31;
  hx = (hx & 0x000fffff) | ((0x3ff - i) << 20);
  y = (double)(k + i);
  x = om_with_hi(x, hx);
  z = y * log10_2lo + ivln10 * om_log(x);
  return z + y * log10_2hi;
}

#endif
