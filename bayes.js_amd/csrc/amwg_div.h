// amwg_div.h -- correctly rounded fp64 division by a loop-invariant divisor.
//
// ld.norm divides every observation's squared residual by the same 2*sd*sd
// (distributions.js:120).  IEEE division on CDNA4 is an 11-instruction sequence with a
// quarter-rate v_rcp_f64; with the divisor b fixed for a whole pass, 1/b is prepared once as
// a double-double  y_hi = RN(1/b) (true IEEE division),  y_lo ~= 1/b - y_hi  (from the exact
// residual 1 - b*y_hi), and each quotient takes four operations:
//     t  = RN(a * y_lo)
//     q1 = RN(a * y_hi + t)          |q1 - a/b| <= (1/2 + 2^-50) ulp: a faithful quotient
//     r  = a - b * q1                exact in one fma (q1 faithful)
//     q  = RN(q1 + r * y_hi)         = RN(a / b)
// The last step is Markstein's correction (IBM J. R&D 34(1), 1990; Muller et al., Handbook of
// Floating-Point Arithmetic, 2nd ed., section 4.7): with q1 faithful and y_hi the correctly
// rounded reciprocal, q is the correctly rounded quotient, provided nothing over/underflows.
// Callers guarantee that range precondition (mid_range) and otherwise use '/', so results are
// bit-identical to IEEE division -- which is what the reference computes.
// tests/test_gpu_math.py compares 4e6 random and adversarial (a,b) pairs with '/' on the device.
#pragma once
#include "amwg_math.h"

namespace amwg {

struct Reciprocal { double hi, lo; };

AMWG_HD Reciprocal make_reciprocal(double b) {
  Reciprocal y;
  y.hi = 1.0 / b;                                 // correctly rounded
  y.lo = __builtin_fma(-b, y.hi, 1.0) * y.hi;     // (1 - b*y_hi) is exact; times ~1/b
  return y;
}

AMWG_HD double div_by_invariant(double a, double b, Reciprocal y) {
  const double t = a * y.lo;
  const double q1 = __builtin_fma(a, y.hi, t);
  const double r = __builtin_fma(-b, q1, a);
  return __builtin_fma(r, y.hi, q1);
}

// true iff 2^-200 <= v <= 2^200 (positive, normal, comfortably inside the exponent range)
AMWG_HD bool mid_range(double v) {
  const uint32_t h = (uint32_t)hi_word(v);
  return (h - 0x33700000u) <= (0x4C700000u - 0x33700000u);
}

// |a| in 2^-600..2^600 (exactly zero is NOT included: 0 takes the IEEE path, it is rare)
AMWG_HD bool wide_range(double a) {
  const uint32_t h = (uint32_t)hi_word(a) & 0x7fffffffu;
  return (h - 0x1A700000u) <= (0x65700000u - 0x1A700000u);
}

}  // namespace amwg
