// amwg_trig.h -- the remaining one- and two-argument Math.* functions of the JavaScript engine the reference runs on, for user
// closures: sin cos tan asin acos atan2 sinh cosh asinh acosh atanh cbrt log2.  Like exp/log/pow in amwg_math.h these are V8's
// algorithms (src/base/ieee754.cc: Sun fdlibm, cbrt/log2 in their FreeBSD msun form), restated operation for operation so that a
// translated closure returns the bits V8 returns; tests/golden/v8_math3_pairs.bin and v8_atan2_pairs.bin (oracle/gen_math3_golden.js,
// 24 000 arguments each incl. huge ones for the Payne-Hanek reduction) pin the host build, tests/test_gpu_math.py the device build.
// Included at the end of amwg_math.h.  The public functions are not force-inlined (AMWG_HD_OUTLINE): a closure that calls Math.sin in five places
// gets one copy, which keeps hiprtc compile times of large closures in seconds.
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
#pragma once

namespace amwg {

AMWG_HD double set_lo_word(double x, uint32_t lo) { return bits_f64((f64_bits(x) & 0xffffffff00000000ull) | (uint64_t)lo); }
AMWG_HD double from_words(uint32_t hi, uint32_t lo) { return bits_f64(((uint64_t)hi << 32) | (uint64_t)lo); }

// ---- argument reduction modulo pi/2 ------------------------------------------------------------------------------------
// 2/pi in 24-bit pieces (1584 bits), for arguments beyond 2^19 * pi/2
AMWG_HD int32_t two_over_pi_word(int i) {
  const int32_t T[66] = {
      0xA2F983, 0x6E4E44, 0x1529FC, 0x2757D1, 0xF534DD, 0xC0DB62, 0x95993C, 0x439041, 0xFE5163, 0xABDEBB, 0xC561B7,
      0x246E3A, 0x424DD2, 0xE00649, 0x2EEA09, 0xD1921C, 0xFE1DEB, 0x1CB129, 0xA73EE8, 0x8235F5, 0x2EBB44, 0x84E99C,
      0x7026B4, 0x5F7E41, 0x3991D6, 0x398353, 0x39F49C, 0x845F8B, 0xBDF928, 0x3B1FF8, 0x97FFDE, 0x05980F, 0xEF2F11,
      0x8B5A0A, 0x6D1F6D, 0x367ECF, 0x27CB09, 0xB74F46, 0x3F669E, 0x5FEA2D, 0x7527BA, 0xC7EBE5, 0xF17B3D, 0x0739F7,
      0x8A5292, 0xEA6BFB, 0x5FB11F, 0x8D5D08, 0x560330, 0x46FC7B, 0x6BABF0, 0xCFBC20, 0x9AF436, 0x1DA9E3, 0x91615E,
      0xE61B08, 0x659985, 0x5F14A0, 0x68408D, 0xFFD880, 0x4D7327, 0x310606, 0x1556CA, 0x73A8C9, 0x60E27B, 0xC08C6B};
  return T[i];
}

// fdlibm k_rem_pio2.c with prec = 2 (53-bit result in two doubles): x[0..nx-1] are 24-bit pieces of |x| scaled by 2^e0
AMWG_HD_OUTLINE int kernel_rem_pio2(const double *x, double *y, int e0, int nx) {
  const double PIo2[8] = {1.57079625129699707031e+00, 7.54978941586159635335e-08, 5.39030252995776476554e-15, 3.28200341580791294123e-22,
                          1.27065575308067607349e-29, 1.22933308981111328932e-36, 2.73370053816464559624e-44, 2.16741683877804819444e-51};
  const double two24 = 1.67772160000000000000e+07, twon24 = 5.96046447753906250000e-08;
  const int jk = 4, jp = 4;
  int32_t iq[20];
  double f[20], fq[20], q[20];
  const int jx = nx - 1;
  int jv = (e0 - 3) / 24;
  if (jv < 0) jv = 0;
  int q0 = e0 - 24 * (jv + 1);
  {
    int j = jv - jx;
    const int m = jx + jk;
    for (int i = 0; i <= m; i++, j++) f[i] = (j < 0) ? 0.0 : (double)two_over_pi_word(j);
  }
  for (int i = 0; i <= jk; i++) {
    double fw = 0.0;
    for (int j = 0; j <= jx; j++) fw += x[j] * f[jx + i - j];
    q[i] = fw;
  }
  int jz = jk, n, ih;
  double z;
  for (;;) {
    // distill q[] into iq[] in reverse order
    z = q[jz];
    for (int i = 0, j = jz; j > 0; i++, j--) {
      const double fw = (double)((int32_t)(twon24 * z));
      iq[i] = (int32_t)(z - two24 * fw);
      z = q[j - 1] + fw;
    }
    z = scalbn_v8(z, q0);
    z -= 8.0 * __builtin_floor(z * 0.125);
    n = (int32_t)z;
    z -= (double)n;
    ih = 0;
    if (q0 > 0) {
      const int32_t i = (iq[jz - 1] >> (24 - q0));
      n += i;
      iq[jz - 1] -= i << (24 - q0);
      ih = iq[jz - 1] >> (23 - q0);
    } else if (q0 == 0) {
      ih = iq[jz - 1] >> 23;
    } else if (z >= 0.5) {
      ih = 2;
    }
    if (ih > 0) {  // q > 0.5
      n += 1;
      int32_t carry = 0;
      for (int i = 0; i < jz; i++) {
        const int32_t j = iq[i];
        if (carry == 0) {
          if (j != 0) { carry = 1; iq[i] = 0x1000000 - j; }
        } else {
          iq[i] = 0xffffff - j;
        }
      }
      if (q0 > 0) {
        if (q0 == 1) iq[jz - 1] &= 0x7fffff;
        else if (q0 == 2) iq[jz - 1] &= 0x3fffff;
      }
      if (ih == 2) {
        z = 1.0 - z;
        if (carry != 0) z -= scalbn_v8(1.0, q0);
      }
    }
    if (z != 0.0) break;
    int32_t j = 0;
    for (int i = jz - 1; i >= jk; i--) j |= iq[i];
    if (j != 0) break;
    int k = 1;
    while (iq[jk - k] == 0) k++;  // k = terms needed
    for (int i = jz + 1; i <= jz + k; i++) {
      f[jx + i] = (double)two_over_pi_word(jv + i);
      double fw = 0.0;
      for (int jj = 0; jj <= jx; jj++) fw += x[jj] * f[jx + i - jj];
      q[i] = fw;
    }
    jz += k;
  }
  if (z == 0.0) {
    jz -= 1;
    q0 -= 24;
    while (iq[jz] == 0) { jz--; q0 -= 24; }
  } else {
    z = scalbn_v8(z, -q0);
    if (z >= two24) {
      const double fw = (double)((int32_t)(twon24 * z));
      iq[jz] = (int32_t)(z - two24 * fw);
      jz += 1;
      q0 += 24;
      iq[jz] = (int32_t)fw;
    } else {
      iq[jz] = (int32_t)z;
    }
  }
  {
    double fw = scalbn_v8(1.0, q0);
    for (int i = jz; i >= 0; i--) { q[i] = fw * (double)iq[i]; fw *= twon24; }
  }
  for (int i = jz; i >= 0; i--) {
    double fw = 0.0;
    for (int k = 0; k <= jp && k <= jz - i; k++) fw += PIo2[k] * q[i + k];
    fq[jz - i] = fw;
  }
  double fw = 0.0;
  for (int i = jz; i >= 0; i--) fw += fq[i];
  y[0] = (ih == 0) ? fw : -fw;
  fw = fq[0] - fw;
  for (int i = 1; i <= jz; i++) fw += fq[i];
  y[1] = (ih == 0) ? fw : -fw;
  return n & 7;
}

// fdlibm e_rem_pio2.c: y[0] + y[1] = x - n*pi/2, |y| <= pi/4; returns n (only its low bits matter to the callers)
AMWG_HD_OUTLINE int rem_pio2_v8(double x, double *y) {
  const double half = 0.5, two24 = 1.67772160000000000000e+07, invpio2 = 6.36619772367581382433e-01,
               pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11, pio2_2 = 6.07710050630396597660e-11,
               pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21, pio2_3t = 8.47842766036889956997e-32;
  const int32_t hx = hi_word(x), ix = hx & 0x7fffffff;
  if (ix <= 0x3fe921fb) { y[0] = x; y[1] = 0; return 0; }
  if (ix < 0x4002d97c) {  // |x| < 3pi/4: n = +-1
    if (hx > 0) {
      double z = x - pio2_1;
      if (ix != 0x3ff921fb) { y[0] = z - pio2_1t; y[1] = (z - y[0]) - pio2_1t; }
      else { z -= pio2_2; y[0] = z - pio2_2t; y[1] = (z - y[0]) - pio2_2t; }
      return 1;
    }
    double z = x + pio2_1;
    if (ix != 0x3ff921fb) { y[0] = z + pio2_1t; y[1] = (z - y[0]) + pio2_1t; }
    else { z += pio2_2; y[0] = z + pio2_2t; y[1] = (z - y[0]) + pio2_2t; }
    return -1;
  }
  if (ix <= 0x413921fb) {  // |x| <= 2^19 * pi/2
    double t = __builtin_fabs(x);
    const int32_t n = (int32_t)(t * invpio2 + half);
    const double fn = (double)n;
    double r = t - fn * pio2_1, w = fn * pio2_1t;
    // npio2_hw[n-1], the high word of n*pi/2: where the first subtraction cancels too much
    bool quick = false;
    if (n < 32) {
      const int32_t npio2_hw[32] = {0x3FF921FB, 0x400921FB, 0x4012D97C, 0x401921FB, 0x401F6A7A, 0x4022D97C, 0x4025FDBB, 0x402921FB, 0x402C463A, 0x402F6A7A, 0x4031475C,
                                    0x4032D97C, 0x40346B9C, 0x4035FDBB, 0x40378FDB, 0x403921FB, 0x403AB41B, 0x403C463A, 0x403DD85A, 0x403F6A7A, 0x40407E4C, 0x4041475C,
                                    0x4042106C, 0x4042D97C, 0x4043A28C, 0x40446B9C, 0x404534AC, 0x4045FDBB, 0x4046C6CB, 0x40478FDB, 0x404858EB, 0x404921FB};
      quick = ix != npio2_hw[n - 1];
    }
    if (quick) {
      y[0] = r - w;
    } else {
      const int32_t j = ix >> 20;
      y[0] = r - w;
      int32_t i = j - ((hi_word(y[0]) >> 20) & 0x7ff);
      if (i > 16) {  // second iteration, good to 118 bits
        t = r;
        w = fn * pio2_2;
        r = t - w;
        w = fn * pio2_2t - ((t - r) - w);
        y[0] = r - w;
        i = j - ((hi_word(y[0]) >> 20) & 0x7ff);
        if (i > 49) {  // third iteration, 151 bits
          t = r;
          w = fn * pio2_3;
          r = t - w;
          w = fn * pio2_3t - ((t - r) - w);
          y[0] = r - w;
        }
      }
    }
    y[1] = (r - y[0]) - w;
    if (hx < 0) { y[0] = -y[0]; y[1] = -y[1]; return -n; }
    return n;
  }
  if (ix >= 0x7ff00000) { y[0] = y[1] = x - x; return 0; }
  // z = scalbn(|x|, -ilogb(x) + 23), cut into three 24-bit pieces
  const int32_t e0 = (ix >> 20) - 1046;
  double z = from_words((uint32_t)(ix - (e0 << 20)), lo_word(x));
  double tx[3];
  for (int i = 0; i < 2; i++) { tx[i] = (double)((int32_t)z); z = (z - tx[i]) * two24; }
  tx[2] = z;
  int nx = 3;
  while (tx[nx - 1] == 0.0) nx--;
  const int n = kernel_rem_pio2(tx, y, e0, nx);
  if (hx < 0) { y[0] = -y[0]; y[1] = -y[1]; return -n; }
  return n;
}

// ---- kernels on [-pi/4, pi/4] ----------------------------------------------------------------------------------------------
AMWG_HD double kernel_sin(double x, double y, int iy) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const int32_t ix = hi_word(x) & 0x7fffffff;
  if (ix < 0x3e400000) { if ((int32_t)x == 0) return x; }
  const double z = x * x, v = z * x;
  const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  if (iy == 0) return x + v * (S1 + z * r);
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

AMWG_HD double kernel_cos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const int32_t ix = hi_word(x) & 0x7fffffff;
  if (ix < 0x3e400000) { if ((int32_t)x == 0) return 1.0; }
  const double z = x * x;
  const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  if (ix < 0x3FD33333) return 1.0 - (0.5 * z - (z * r - x * y));
  const double qx = (ix > 0x3fe90000) ? 0.28125 : from_words((uint32_t)(ix - 0x00200000), 0u);   // x/4
  const double hz = 0.5 * z - qx, a = 1.0 - qx;
  return a - (hz - (z * r - x * y));
}

AMWG_HD double kernel_tan(double x, double y, int iy) {
  const double T0 = 3.33333333333334091986e-01, T1 = 1.33333333333201242699e-01, T2 = 5.39682539762260521377e-02, T3 = 2.18694882948595424599e-02,
               T4 = 8.86323982359930005737e-03, T5 = 3.59207910759131235356e-03, T6 = 1.45620945432529025516e-03, T7 = 5.88041240820264096874e-04,
               T8 = 2.46463134818469906812e-04, T9 = 7.81794442939557092300e-05, T10 = 7.14072491382608190305e-05, T11 = -1.85586374855275456654e-05,
               T12 = 2.59073051863633712884e-05;
  const double pio4 = 7.85398163397448278999e-01, pio4lo = 3.06161699786838301793e-17;
  const int32_t hx = hi_word(x), ix = hx & 0x7fffffff;
  double z, r, v, w, s;
  if (ix < 0x3e300000) {  // |x| < 2^-28
    if ((int32_t)x == 0) {
      if (((ix | (int32_t)lo_word(x)) | (iy + 1)) == 0) return 1.0 / __builtin_fabs(x);
      if (iy == 1) return x;
      // -1 / (x + y), carefully
      z = w = x + y;
      z = set_lo_word(z, 0u);
      v = y - (z - x);
      double a = -1.0 / w, t = a;
      t = set_lo_word(t, 0u);
      s = 1.0 + t * z;
      return t + a * (s + t * v);
    }
  }
  if (ix >= 0x3FE59428) {  // |x| >= 0.6744
    if (hx < 0) { x = -x; y = -y; }
    z = pio4 - x;
    w = pio4lo - y;
    x = z + w;
    y = 0.0;
  }
  z = x * x;
  w = z * z;
  // odd and even parts of the polynomial separately
  r = T1 + w * (T3 + w * (T5 + w * (T7 + w * (T9 + w * T11))));
  v = z * (T2 + w * (T4 + w * (T6 + w * (T8 + w * (T10 + w * T12)))));
  s = z * x;
  r = y + z * (s * (r + v) + y);
  r += T0 * s;
  w = x + r;
  if (ix >= 0x3FE59428) {
    v = (double)iy;
    return (double)(1 - ((hx >> 30) & 2)) * (v - 2.0 * (x - (w * w / (w + v) - r)));
  }
  if (iy == 1) return w;
  // -1 / (x + r), accurately
  z = w;
  z = set_lo_word(z, 0u);
  v = r - (z - x);
  double a = -1.0 / w, t = a;
  t = set_lo_word(t, 0u);
  s = 1.0 + t * z;
  return t + a * (s + t * v);
}

AMWG_HD_OUTLINE double sin_v8(double x) {
  const int32_t ix = hi_word(x) & 0x7fffffff;
  if (ix <= 0x3fe921fb) return kernel_sin(x, 0.0, 0);
  if (ix >= 0x7ff00000) return x - x;
  double y[2];
  const int n = rem_pio2_v8(x, y);
  switch (n & 3) {
    case 0: return kernel_sin(y[0], y[1], 1);
    case 1: return kernel_cos(y[0], y[1]);
    case 2: return -kernel_sin(y[0], y[1], 1);
    default: return -kernel_cos(y[0], y[1]);
  }
}

AMWG_HD_OUTLINE double cos_v8(double x) {
  const int32_t ix = hi_word(x) & 0x7fffffff;
  if (ix <= 0x3fe921fb) return kernel_cos(x, 0.0);
  if (ix >= 0x7ff00000) return x - x;
  double y[2];
  const int n = rem_pio2_v8(x, y);
  switch (n & 3) {
    case 0: return kernel_cos(y[0], y[1]);
    case 1: return -kernel_sin(y[0], y[1], 1);
    case 2: return -kernel_cos(y[0], y[1]);
    default: return kernel_sin(y[0], y[1], 1);
  }
}

AMWG_HD_OUTLINE double tan_v8(double x) {
  const int32_t ix = hi_word(x) & 0x7fffffff;
  if (ix <= 0x3fe921fb) return kernel_tan(x, 0.0, 1);
  if (ix >= 0x7ff00000) return x - x;
  double y[2];
  const int n = rem_pio2_v8(x, y);
  return kernel_tan(y[0], y[1], 1 - ((n & 1) << 1));   // 1: tan, -1: -1/tan
}

// ---- inverse trigonometric ---------------------------------------------------------------------------------------------------
AMWG_HD double asin_acos_ratio(double t) {   // p(t)/q(t) of e_asin.c / e_acos.c
  const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
               pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05, qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
               qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
  const double p = t * (pS0 + t * (pS1 + t * (pS2 + t * (pS3 + t * (pS4 + t * pS5)))));
  const double q = 1.0 + t * (qS1 + t * (qS2 + t * (qS3 + t * qS4)));
  return p / q;
}

AMWG_HD_OUTLINE double asin_v8(double x) {
  const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, pio4_hi = 7.85398163397448278999e-01;
  const int32_t hx = hi_word(x), ix = hx & 0x7fffffff;
  if (ix >= 0x3ff00000) {  // |x| >= 1
    if (((ix - 0x3ff00000) | (int32_t)lo_word(x)) == 0) return x * pio2_hi + x * pio2_lo;
    return (x - x) / (x - x);
  }
  if (ix < 0x3fe00000) {  // |x| < 0.5
    if (ix < 0x3e400000) return x;
    const double t = x * x;
    return x + x * asin_acos_ratio(t);
  }
  double w = 1.0 - __builtin_fabs(x);
  double t = w * 0.5;
  const double ratio = asin_acos_ratio(t);
  const double s = __builtin_sqrt(t);
  if (ix >= 0x3FEF3333) {  // |x| > 0.975
    t = pio2_hi - (2.0 * (s + s * ratio) - pio2_lo);
  } else {
    w = set_lo_word(s, 0u);
    const double c = (t - w * w) / (s + w);
    const double p = 2.0 * s * ratio - (pio2_lo - 2.0 * c);
    const double q = pio4_hi - 2.0 * w;
    t = pio4_hi - (p - q);
  }
  return hx > 0 ? t : -t;
}

AMWG_HD_OUTLINE double acos_v8(double x) {
  const double pi = 3.14159265358979311600e+00, pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
  const int32_t hx = hi_word(x), ix = hx & 0x7fffffff;
  if (ix >= 0x3ff00000) {
    if (((ix - 0x3ff00000) | (int32_t)lo_word(x)) == 0) return hx > 0 ? 0.0 : pi + 2.0 * pio2_lo;
    return (x - x) / (x - x);
  }
  if (ix < 0x3fe00000) {  // |x| < 0.5
    if (ix <= 0x3c600000) return pio2_hi + pio2_lo;
    const double z = x * x;
    const double r = asin_acos_ratio(z);
    return pio2_hi - (x - (pio2_lo - x * r));
  }
  if (hx < 0) {  // x < -0.5
    const double z = (1.0 + x) * 0.5;
    const double r = asin_acos_ratio(z);
    const double s = __builtin_sqrt(z);
    const double w = r * s - pio2_lo;
    return pi - 2.0 * (s + w);
  }
  const double z = (1.0 - x) * 0.5;
  const double s = __builtin_sqrt(z);
  const double df = set_lo_word(s, 0u);
  const double c = (z - df * df) / (s + df);
  const double r = asin_acos_ratio(z);
  const double w = r * s + c;
  return 2.0 * (df + w);
}

AMWG_HD_OUTLINE double atan2_v8(double y, double x) {
  const double tiny = 1.0e-300, pi_o_4 = 7.8539816339744827900E-01, pi_o_2 = 1.5707963267948965580E+00, pi = 3.1415926535897931160E+00,
               pi_lo = 1.2246467991473531772E-16;
  const int32_t hx = hi_word(x), hy = hi_word(y);
  const uint32_t lx = lo_word(x), ly = lo_word(y);
  const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  if (((uint32_t)ix | ((lx | (0u - lx)) >> 31)) > 0x7ff00000u || ((uint32_t)iy | ((ly | (0u - ly)) >> 31)) > 0x7ff00000u) return x + y;   // NaN
  if (((hx - 0x3ff00000) | (int32_t)lx) == 0) return atan_v8(y);   // x = 1
  int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2);                  // 2*sign(x) + sign(y)
  if ((iy | (int32_t)ly) == 0) {                                    // y = 0
    switch (m) {
      case 0: case 1: return y;
      case 2: return pi + tiny;
      default: return -pi - tiny;
    }
  }
  if ((ix | (int32_t)lx) == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;   // x = 0
  if (ix == 0x7ff00000) {                                           // x = +-inf
    if (iy == 0x7ff00000) {
      switch (m) {
        case 0: return pi_o_4 + tiny;
        case 1: return -pi_o_4 - tiny;
        case 2: return 3.0 * pi_o_4 + tiny;
        default: return -3.0 * pi_o_4 - tiny;
      }
    }
    switch (m) {
      case 0: return 0.0;
      case 1: return -0.0;
      case 2: return pi + tiny;
      default: return -pi - tiny;
    }
  }
  if (iy == 0x7ff00000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;   // y = +-inf
  const int32_t k = (iy - ix) >> 20;
  double z;
  if (k > 60) { z = pi_o_2 + 0.5 * pi_lo; m &= 1; }      // |y/x| > 2^60
  else if (hx < 0 && k < -60) z = 0.0;                    // 0 > |y|/x > -2^-60
  else z = atan_v8(__builtin_fabs(y / x));
  switch (m) {
    case 0: return z;
    case 1: return -z;
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}

// ---- hyperbolic ------------------------------------------------------------------------------------------------------------------
AMWG_HD_OUTLINE double sinh_v8(double x) {
  const double KSINH_OVERFLOW = 710.4758600739439, TWO_M28 = 3.725290298461914e-9, LOG_MAXD = 709.7822265625, shuge = 1.0e307;
  const double h = (x < 0) ? -0.5 : 0.5;
  const double ax = __builtin_fabs(x);
  if (ax < 22) {
    if (ax < TWO_M28) return x;
    const double t = expm1_v8(ax);
    if (ax < 1) return h * (2 * t - t * t / (t + 1));
    return h * (t + t / (t + 1));
  }
  if (ax < LOG_MAXD) return h * exp_v8(ax);
  if (ax <= KSINH_OVERFLOW) {
    const double w = exp_v8(0.5 * ax);
    const double t = h * w;
    return t * w;
  }
  return x * shuge;   // overflow, inf or NaN
}

AMWG_HD_OUTLINE double cosh_v8(double x) {
  const double KCOSH_OVERFLOW = 710.4758600739439, huge = 1.0e+300;
  const int32_t ix = hi_word(x) & 0x7fffffff;
  if (ix < 0x3fd62e43) {  // |x| < 0.5 ln 2
    const double t = expm1_v8(__builtin_fabs(x));
    const double w = 1.0 + t;
    if (ix < 0x3c800000) return w;
    return 1.0 + (t * t) / (w + w);
  }
  if (ix < 0x40360000) {  // |x| < 22
    const double t = exp_v8(__builtin_fabs(x));
    return 0.5 * t + 0.5 / t;
  }
  if (ix < 0x40862e42) return 0.5 * exp_v8(__builtin_fabs(x));
  if (__builtin_fabs(x) <= KCOSH_OVERFLOW) {
    const double w = exp_v8(0.5 * __builtin_fabs(x));
    const double t = 0.5 * w;
    return t * w;
  }
  if (ix >= 0x7ff00000) return x * x;
  return huge * huge;
}

AMWG_HD_OUTLINE double asinh_v8(double x) {
  const double ln2 = 6.93147180559945286227e-01;
  const int32_t hx = hi_word(x), ix = hx & 0x7fffffff;
  if (ix >= 0x7ff00000) return x + x;
  if (ix < 0x3e300000) return x;   // |x| < 2^-28
  double w;
  if (ix > 0x41b00000) {           // |x| > 2^28
    w = log_v8(__builtin_fabs(x)) + ln2;
  } else if (ix > 0x40000000) {    // 2 < |x| <= 2^28
    const double t = __builtin_fabs(x);
    w = log_v8(2.0 * t + 1.0 / (__builtin_sqrt(x * x + 1.0) + t));
  } else {
    const double t = x * x;
    w = log1p_v8(__builtin_fabs(x) + t / (1.0 + __builtin_sqrt(1.0 + t)));
  }
  return hx > 0 ? w : -w;
}

AMWG_HD_OUTLINE double acosh_v8(double x) {
  const double ln2 = 6.93147180559945286227e-01;
  const int32_t hx = hi_word(x);
  if (hx < 0x3ff00000) return (x - x) / (x - x);   // x < 1
  if (hx >= 0x41b00000) {                           // x > 2^28
    if (hx >= 0x7ff00000) return x + x;
    return log_v8(x) + ln2;
  }
  if (((hx - 0x3ff00000) | (int32_t)lo_word(x)) == 0) return 0.0;
  if (hx > 0x40000000) {                            // 2 < x < 2^28
    const double t = x * x;
    return log_v8(2.0 * x - 1.0 / (x + __builtin_sqrt(t - 1.0)));
  }
  const double t = x - 1.0;
  return log1p_v8(t + __builtin_sqrt(2.0 * t + t * t));
}

AMWG_HD_OUTLINE double atanh_v8(double x) {
  const int32_t hx = hi_word(x), ix = hx & 0x7fffffff;
  const uint32_t lx = lo_word(x);
  if (((uint32_t)ix | ((lx | (0u - lx)) >> 31)) > 0x3ff00000u) return (x - x) / (x - x);   // |x| > 1
  if (ix == 0x3ff00000) return x / 0.0;
  if (ix < 0x3e300000) return x;   // |x| < 2^-28
  x = set_hi_word(x, ix);           // |x|
  double t;
  if (ix < 0x3fe00000) {            // |x| < 0.5
    t = x + x;
    t = 0.5 * log1p_v8(t + t * x / (1.0 - x));
  } else {
    t = 0.5 * log1p_v8((x + x) / (1.0 - x));
  }
  return hx >= 0 ? t : -t;
}

// ---- cbrt, log2 (FreeBSD msun s_cbrt.c / e_log2.c, as V8 carries them) ---------------------------------------------------------------
AMWG_HD_OUTLINE double cbrt_v8(double x) {
  const uint32_t B1 = 715094163u, B2 = 696219795u;
  const double P0 = 1.87595182427177009643, P1 = -1.88497979543377169875, P2 = 1.621429720105354466140, P3 = -0.758397934778766047437,
               P4 = 0.145996192886612446982;
  const uint32_t hx0 = (uint32_t)hi_word(x);
  const uint32_t sign = hx0 & 0x80000000u, hx = hx0 ^ sign;
  if (hx >= 0x7ff00000u) return x + x;
  double t;
  if (hx < 0x00100000u) {  // zero or subnormal
    if ((hx | lo_word(x)) == 0) return x;
    t = from_words(0x43500000u, 0u) * x;   // 2^54 * x
    const uint32_t high = (uint32_t)hi_word(t);
    t = from_words(sign | ((high & 0x7fffffffu) / 3 + B2), 0u);
  } else {
    t = from_words(sign | (hx / 3 + B1), 0u);
  }
  // new cbrt to 23 bits
  double r = (t * t) * (t / x);
  t = t * ((P0 + r * (P1 + r * P2)) + ((r * r) * r) * (P3 + r * P4));
  // round t away from zero to 23 bits
  t = bits_f64((f64_bits(t) + 0x80000000ull) & 0xffffffffc0000000ull);
  // one Newton step to 53 bits
  const double s = t * t;
  r = x / s;
  const double w = t + t;
  r = (r - t) / (w + r);
  return t + t * r;
}

AMWG_HD_OUTLINE double log2_v8(double x) {
  const double two54 = 1.80143985094819840000e+16, ivln2hi = 1.44269504072144627571e+00, ivln2lo = 1.67517131648865118353e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
               Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01, Lg7 = 1.479819860511658591e-01;
  int32_t hx = hi_word(x), k = 0;
  const uint32_t lx = lo_word(x);
  if (hx < 0x00100000) {  // x < 2^-1022
    if (((hx & 0x7fffffff) | (int32_t)lx) == 0) return -two54 / 0.0;
    if (hx < 0) return (x - x) / 0.0;
    k -= 54;
    x *= two54;
    hx = hi_word(x);
  }
  if (hx >= 0x7ff00000) return x + x;
  if (hx == 0x3ff00000 && lx == 0) return 0.0;   // log2(1) = +0
  k += (hx >> 20) - 1023;
  hx &= 0x000fffff;
  const int32_t i = (hx + 0x95f64) & 0x100000;
  x = set_hi_word(x, hx | (i ^ 0x3ff00000));   // normalize x or x/2
  k += (i >> 20);
  const double y = (double)k;
  const double f = x - 1.0;
  const double hfsq = 0.5 * f * f;
  // k_log1p(f)
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double w4 = z * z;
  const double t1 = w4 * (Lg2 + w4 * (Lg4 + w4 * Lg6));
  const double t2 = z * (Lg1 + w4 * (Lg3 + w4 * (Lg5 + w4 * Lg7)));
  const double r = s * (hfsq + (t2 + t1));
  double hi = f - hfsq;
  hi = set_lo_word(hi, 0u);
  const double lo = (f - hi) - hfsq + r;
  double val_hi = hi * ivln2hi;
  double val_lo = (lo + hi) * ivln2lo + lo * ivln2hi;
  const double w = y + val_hi;
  val_lo += (y - w) + val_hi;
  val_hi = w;
  return val_lo + val_hi;
}

// ---- Math.hypot (V8 builtins-math.cc: largest magnitude factored out, Kahan-compensated sum of the squared ratios) -------------------------
AMWG_HD_OUTLINE double hypot_v8(const double *v, int n) {
  bool any_nan = false;
  double mx = 0.0;
  for (int i = 0; i < n; i++) {
    const double a = __builtin_fabs(v[i]);
    if (a != a) any_nan = true;
    else if (a > mx) mx = a;
  }
  if (mx == __builtin_inf()) return mx;      // an infinite argument wins over NaN
  if (any_nan) return __builtin_nan("");
  if (mx == 0.0) return 0.0;
  double sum = 0.0, compensation = 0.0;
  for (int i = 0; i < n; i++) {
    const double r = __builtin_fabs(v[i]) / mx;
    const double summand = (r * r) - compensation;
    const double preliminary = sum + summand;
    compensation = (preliminary - sum) - summand;
    sum = preliminary;
  }
  return __builtin_sqrt(sum) * mx;
}
AMWG_HD double hypot2_v8(double a, double b) { const double v[2] = {a, b}; return hypot_v8(v, 2); }
AMWG_HD double hypot3_v8(double a, double b, double c) { const double v[3] = {a, b, c}; return hypot_v8(v, 3); }
AMWG_HD double hypot4_v8(double a, double b, double c, double d) { const double v[4] = {a, b, c, d}; return hypot_v8(v, 4); }

}  // namespace amwg
