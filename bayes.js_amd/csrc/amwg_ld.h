// amwg_ld.h -- the ld.* log densities on the path, as device functions with the
// reference's expression trees (fp64, one rounding per operation).
//   lgamma / lfactorial / lbeta   distributions.js:63-77, 79-82, 89-92
//   ld.beta   :104-113     ld.norm  :119-121     ld.unif  :221-223
//   ld.bern   :228-230     ld.pois  :282-284
//   cauchy, laplace/dexp, gamma, invgamma, lnorm, pareto, t, weibull, logis, exp, binom, nbinom,
//   hyper, lchoose            distributions.js:84-86, 115-280 (cited per function below)
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
// the same two with LITERAL parameters, their constants folded by the translator (translate.js foldConstantDensities: c = -0.5 log(2 pi) - log(sd) and den = (2 sd) sd;
// log(1 / (hi - lo))) -- the same expression trees, the logarithms taken once at translation time by the V8 whose Math.log log_v8 restates
AMWG_HD double ld_norm_c(double x, double mean, double c, double den) {
  const double t = x - mean;
  return c - (t * t) / den;
}
AMWG_HD double ld_unif_c(double x, double lo, double hi, double log_inv_width) { return (x < lo || x > hi) ? -kInf : log_inv_width; }
AMWG_HD double ld_beta(double x, double a, double b) {
  if (x > 1 || x < 0) return -kInf;
  if (a == 1 && b == 1) return 0;
  return (a - 1) * log_v8(x) + (b - 1) * log_v8(1 - x) - lbeta_js(a, b);
}
AMWG_HD double ld_bern(double x, double p) { return !(x == 0 || x == 1) ? -kInf : log_v8(x * p + (1 - x) * (1 - p)); }
AMWG_HD double ld_pois(double x, double lambda) { return x < 0 ? -kInf : log_v8(lambda) * x - lambda - lfactorial_js(x); }

// ---- the remaining scalar densities of distributions.js, same expression trees.
// pow(t, 2) is t*t (V8's pow special case), general exponents go through pow_v8.
AMWG_HD double lchoose_js(double n, double k) { return lfactorial_js(n) - lfactorial_js(k) - lfactorial_js(n - k); }   // :84-86
AMWG_HD double ld_cauchy(double x, double location, double scale) {   // :115-117
  const double t = x - location;
  return log_v8(scale) - log_v8(t * t + scale * scale) - log_v8(kPi);
}
AMWG_HD double ld_laplace(double x, double location, double scale) {  // :136-138 (ld.dexp is the same function)
  return (-__builtin_fabs(x - location) / scale) - log_v8(2 * scale);
}
AMWG_HD double ld_gamma(double x, double shape, double rate) {   // :142-152
  const double scale = 1 / rate;
  if (x < 0) return -kInf;
  if (x == 0 && shape == 1) return -log_v8(scale);
  return (shape - 1) * log_v8(x) - x / scale - lgamma_js(shape) - shape * log_v8(scale);
}
AMWG_HD double ld_invgamma(double x, double shape, double scale) {   // :154-159
  if (x <= 0) return -kInf;
  return -(shape + 1) * log_v8(x) - scale / x - lgamma_js(shape) + shape * log_v8(scale);
}
AMWG_HD double ld_lnorm(double x, double meanlog, double sdlog) {   // :161-167
  if (x <= 0) return -kInf;
  const double t = log_v8(x) - meanlog;
  return -log_v8(x) - 0.5 * log_v8(2 * kPi) - log_v8(sdlog) - (t * t) / (2 * sdlog * sdlog);
}
AMWG_HD double ld_pareto(double x, double scale, double shape) {   // :169-174
  if (x < scale) return -kInf;
  return log_v8(shape) + shape * log_v8(scale) - (shape + 1) * log_v8(x);
}
AMWG_HD double ld_t(double x, double location, double scale, double df) {   // :176-180
  df = df > 1e100 ? 1e100 : df;
  const double q = (x - location) / scale;
  return lgamma_js((df + 1) / 2) - lgamma_js(df / 2) - log_v8(__builtin_sqrt(kPi * df) * scale) +
         log_v8(pow_v8(1 + (1 / df) * (q * q), -(df + 1) / 2));
}
AMWG_HD double ld_weibull(double x, double shape, double scale) {   // :185-191
  if (x < 0) return -kInf;
  if (x == 0 && shape < 1) return kInf;
  const double tmp1 = pow_v8(x / scale, shape - 1);
  const double tmp2 = tmp1 * (x / scale);
  return -tmp2 + log_v8(shape * tmp1 / scale);
}
AMWG_HD double ld_logis(double x, double location, double scale) {   // :196-201
  x = __builtin_fabs((x - location) / scale);
  const double e = exp_v8(-x);
  const double f = 1.0 + e;
  return -(x + log_v8(scale * f * f));
}
AMWG_HD double ld_exp(double x, double rate) { return x < 0 ? -kInf : log_v8(rate) - rate * x; }   // :217-219
AMWG_HD double ld_binom(double x, double size, double prob) {   // :240-248
  if (x > size || x < 0) return -kInf;
  if (prob == 0 || prob == 1) return (size * prob) == x ? 0.0 : -kInf;
  return lchoose_js(size, x) + x * log_v8(prob) + (size - x) * log_v8(1 - prob);
}
AMWG_HD double ld_nbinom(double x, double size, double prob) {   // :267-272
  if (x < 0) return -kInf;
  return lchoose_js(x + size - 1, size - 1) + x * log_v8(1 - prob) + size * log_v8(prob);
}
AMWG_HD double ld_hyper(double x, double m, double n, double k) {   // :274-280
  if (x < 0 || x > k) return -kInf;
  return lchoose_js(m, x) + lchoose_js(n, k - x) - lchoose_js(m + n, k);
}

// ld.norm split into its loop-invariant part and its per-observation part: for a whole pass
// over the data (mean, sd) are fixed, so   ld.norm(x) = c - (x-mean)^2 / den   with
//   c = (-0.5*log(2*pi)) - log(sd)        den = (2*sd)*sd
// evaluated once -- the same roundings, in the same order, as the full expression.
AMWG_HD double norm_c(double neg_half_log_2pi, double sd) { return neg_half_log_2pi - log_v8(sd); }
AMWG_HD double norm_den(double sd) { return 2 * sd * sd; }

}  // namespace amwg
