'use strict';
/*
 * ld.js -- host-side (JavaScript) versions of the log densities that are on the GPU path, so
 * a user's log_post(state, data) closure written against the reference's `ld` object
 * (distributions.js) still runs on the host, e.g. to evaluate a density by hand.  The sampler
 * itself never calls these: it runs the HIP twins in csrc/amwg_ld.h.
 * Parameterisation and expression order follow distributions.js:63-92 (lgamma, lfactorial, lchoose,
 * lbeta), :104-113 (beta), :115-117 (cauchy), :119-121 (norm), :125-134 (bivarnorm), :136-140
 * (laplace/dexp), :142-159 (gamma, invgamma), :161-174 (lnorm, pareto), :176-180 (t), :185-201
 * (weibull, logis), :203-214 (dirichlet), :217-223 (exp, unif), :228-238 (bern, cat), :240-248
 * (binom), :267-284 (nbinom, hyper, pois).  tests/js/test_frontend.js compares every function
 * with the reference's on the committed golden arguments.
 */
const LANCZOS = [76.18009172947146, -86.50532032941677, 24.01409824083091,
  -1.231739572450155, 0.1208650973866179e-2, -0.5395239384953e-5];

function lgamma(x) {
  let y = x, t = x + 5.5, ser = 1.000000000190015;
  t -= (x + 0.5) * Math.log(t);
  for (let j = 0; j < 6; j++) ser += LANCZOS[j] / ++y;
  return Math.log(2.5066282746310005 * ser / x) - t;
}
const lfactorial = (n) => (n < 0 ? NaN : lgamma(n + 1));
const lchoose = (n, k) => lfactorial(n) - lfactorial(k) - lfactorial(n - k);
const lbeta = (a, b) => lgamma(a) + lgamma(b) - lgamma(a + b);
const log = Math.log, exp = Math.exp, abs = Math.abs, pow = Math.pow, sqrt = Math.sqrt, pi = Math.PI;

const ld = {
  lgamma, lfactorial, lchoose, lbeta,
  cauchy(x, location, scale) { return log(scale) - log(pow(x - location, 2) + pow(scale, 2)) - log(pi); },
  bivarnorm(x, mean, sd, corr) {
    const z = pow(x[0] - mean[0], 2) / pow(sd[0], 2) + pow(x[1] - mean[1], 2) / pow(sd[1], 2) -
              (2 * corr * (x[0] - mean[0]) * (x[1] - mean[1])) / (sd[0] * sd[1]);
    const normalizing_factor = -(log(2) + log(pi) + log(sd[0]) + log(sd[1]) + 0.5 * log(1 - pow(corr, 2)));
    return normalizing_factor - z / (2 * (1 - pow(corr, 2)));
  },
  laplace(x, location, scale) { return (-abs(x - location) / scale) - log(2 * scale); },
  gamma(x, shape, rate) {
    const scale = 1 / rate;
    if (x < 0) return -Infinity;
    if (x === 0 && shape === 1) return -log(scale);
    return (shape - 1) * log(x) - x / scale - lgamma(shape) - shape * log(scale);
  },
  invgamma(x, shape, scale) { return x <= 0 ? -Infinity : -(shape + 1) * log(x) - scale / x - lgamma(shape) + shape * log(scale); },
  lnorm(x, meanlog, sdlog) {
    if (x <= 0) return -Infinity;
    return -log(x) - 0.5 * log(2 * pi) - log(sdlog) - pow(log(x) - meanlog, 2) / (2 * sdlog * sdlog);
  },
  pareto(x, scale, shape) { return x < scale ? -Infinity : log(shape) + shape * log(scale) - (shape + 1) * log(x); },
  t(x, location, scale, df) {
    df = df > 1e100 ? 1e100 : df;
    return lgamma((df + 1) / 2) - lgamma(df / 2) - log(sqrt(pi * df) * scale) +
           log(pow(1 + (1 / df) * pow((x - location) / scale, 2), -(df + 1) / 2));
  },
  weibull(x, shape, scale) {
    if (x < 0) return -Infinity;
    if (x === 0 && shape < 1) return Infinity;
    const tmp1 = pow(x / scale, shape - 1);
    const tmp2 = tmp1 * (x / scale);
    return -tmp2 + log(shape * tmp1 / scale);
  },
  logis(x, location, scale) {
    x = abs((x - location) / scale);
    const e = exp(-x);
    const f = 1.0 + e;
    return -(x + log(scale * f * f));
  },
  dirichlet(x, alpha) {
    let sum_alpha = 0, sum_lgamma_alpha = 0, sum_alpha_sub_1_log_x = 0;
    for (let i = 0; i < alpha.length; i++) {
      sum_alpha += alpha[i];
      sum_lgamma_alpha += lgamma(alpha[i]);
      sum_alpha_sub_1_log_x += (alpha[i] - 1) * log(x[i]);
    }
    return lgamma(sum_alpha) - sum_lgamma_alpha + sum_alpha_sub_1_log_x;
  },
  exp(x, rate) { return x < 0 ? -Infinity : log(rate) - rate * x; },
  cat(x, probs) { return (x < 1 || x > probs.length) ? -Infinity : log(probs[x - 1]); },
  binom(x, size, prob) {
    if (x > size || x < 0) return -Infinity;
    if (prob === 0 || prob === 1) return (size * prob) === x ? 0 : -Infinity;
    return lchoose(size, x) + x * log(prob) + (size - x) * log(1 - prob);
  },
  nbinom(x, size, prob) { return x < 0 ? -Infinity : lchoose(x + size - 1, size - 1) + x * log(1 - prob) + size * log(prob); },
  hyper(x, m, n, k) { return (x < 0 || x > k) ? -Infinity : lchoose(m, x) + lchoose(n, k - x) - lchoose(m + n, k); },
  norm(x, mean, sd) { return -0.5 * Math.log(2 * Math.PI) - Math.log(sd) - Math.pow(x - mean, 2) / (2 * sd * sd); },
  unif(x, min, max) { return (x < min || x > max) ? -Infinity : Math.log(1 / (max - min)); },
  beta(x, a, b) {
    if (x > 1 || x < 0) return -Infinity;
    if (a === 1 && b === 1) return 0;
    return (a - 1) * Math.log(x) + (b - 1) * Math.log(1 - x) - lbeta(a, b);
  },
  bern(x, p) { return !(x === 0 || x === 1) ? -Infinity : Math.log(x * p + (1 - x) * (1 - p)); },
  pois(x, lambda) { return x < 0 ? -Infinity : Math.log(lambda) * x - lambda - lfactorial(x); },
};
ld.dexp = ld.laplace;
module.exports = ld;
