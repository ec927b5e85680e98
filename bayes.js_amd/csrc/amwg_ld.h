// amwg_ld.h -- the ld.* log densities on the path, as device functions with the
// reference's expression trees (fp64, one rounding per operation).
//   lgamma / lfactorial / lbeta   distributions.js:63-77, 79-82, 89-92
//   ld.beta   :104-113     ld.norm  :119-121     ld.unif  :221-223
//   ld.bern   :228-230     ld.pois  :282-284
// Math.pow(t, 2) is evaluated as t*t: V8 returns exactly that for exponent 2 (pinned by the
// golden trajectories, which would diverge otherwise).
#pragma once
#include "amwg_math.h"

namespace amwg {

constexpr double kPi = 3.141592653589793;
constexpr double kInf = __builtin_huge_val();

AMWG_HD double lgamma_js(double x) {
  const double cof[6] = {76.18009172947146,  -86.50532032941677,    24.01409824083091,
                         -1.231739572450155, 0.1208650973866179e-2, -0.5395239384953e-5};
  double ser = 1.000000000190015, y = x, tmp = x + 5.5;
  tmp -= (x + 0.5) * log_v8(tmp);
#pragma unroll
  for (int j = 0; j < 6; ++j) { y += 1.0; ser += cof[j] / y; }
  return log_v8(2.5066282746310005 * ser / x) - tmp;
}
AMWG_HD double lfactorial_js(double n) { return n < 0 ? __builtin_nan("") : lgamma_js(n + 1); }
AMWG_HD double lbeta_js(double a, double b) { return lgamma_js(a) + lgamma_js(b) - lgamma_js(a + b); }

AMWG_HD double ld_norm(double x, double mean, double sd) {
  const double t = x - mean;
  return -0.5 * log_v8(2 * kPi) - log_v8(sd) - (t * t) / (2 * sd * sd);
}
AMWG_HD double ld_unif(double x, double lo, double hi) { return (x < lo || x > hi) ? -kInf : log_v8(1 / (hi - lo)); }
AMWG_HD double ld_beta(double x, double a, double b) {
  if (x > 1 || x < 0) return -kInf;
  if (a == 1 && b == 1) return 0;
  return (a - 1) * log_v8(x) + (b - 1) * log_v8(1 - x) - lbeta_js(a, b);
}
AMWG_HD double ld_bern(double x, double p) { return !(x == 0 || x == 1) ? -kInf : log_v8(x * p + (1 - x) * (1 - p)); }
AMWG_HD double ld_pois(double x, double lambda) { return x < 0 ? -kInf : log_v8(lambda) * x - lambda - lfactorial_js(x); }

// ld.norm split into its loop-invariant part and its per-observation part: for a whole pass
// over the data (mean, sd) are fixed, so   ld.norm(x) = c - (x-mean)^2 / den   with
//   c = (-0.5*log(2*pi)) - log(sd)        den = (2*sd)*sd
// evaluated once -- the same roundings, in the same order, as the full expression.
AMWG_HD double norm_c(double neg_half_log_2pi, double sd) { return neg_half_log_2pi - log_v8(sd); }
AMWG_HD double norm_den(double sd) { return 2 * sd * sd; }

}  // namespace amwg
