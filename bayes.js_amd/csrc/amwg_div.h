// amwg_div.h -- correctly rounded fp64 division by a loop-invariant divisor.
//
// ld.norm divides every observation's squared residual by the same 2*sd*sd
// (distributions.js:120).  IEEE division on CDNA4 is an 11-instruction sequence with a
// quarter-rate v_rcp_f64; with the divisor fixed for a whole pass the reciprocal
// y = RN(1/b) is computed once (true IEEE division) and each quotient is
//     q0 = RN(a*y);  r0 = RN(a - b*q0);  q1 = RN(q0 + r0*y);  r1 = a - b*q1 (exact);  q = RN(q1 + r1*y)
// After the first correction q1 is a faithful quotient, and Markstein's theorem (IBM J. R&D
// 34(1), 1990; Muller et al., Handbook of Floating-Point Arithmetic §4.7) then gives
// q = RN(a/b) exactly, provided y is the correctly rounded reciprocal and nothing
// over/underflows.  Callers guarantee the range precondition (see div_range_ok) and fall
// back to '/' otherwise, so results are bit-identical to IEEE division -- which is what the
// reference computes.  tests/test_gpu_math.py compares 4e6 random and adversarial (a,b)
// pairs against '/' on the device.
#pragma once
#include "amwg_math.h"

namespace amwg {

AMWG_HD double div_by_invariant(double a, double b, double y) {
  double q = a * y;
  double r = __builtin_fma(-b, q, a);
  q = __builtin_fma(r, y, q);
  r = __builtin_fma(-b, q, a);
  return __builtin_fma(r, y, q);
}

// true iff 2^-200 <= v <= 2^200 (positive, normal, comfortably inside the exponent range)
AMWG_HD bool mid_range(double v) {
  const uint32_t h = (uint32_t)hi_word(v);
  return (h - 0x33700000u) <= (0x4C700000u - 0x33700000u);
}

}  // namespace amwg
