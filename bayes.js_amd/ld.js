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
const L = Math.log, NEG = -Infinity, PI = Math.PI;
const sq = (t) => t * t;          // V8 evaluates Math.pow(t, 2) as exactly t*t (pinned by the goldens)

// Lanczos approximation with g = 5 and six coefficients (Numerical Recipes gammln), as the reference carries it
function lgamma(x) {
  let y = x, t = x + 5.5, ser = 1.000000000190015;
  t -= (x + 0.5) * L(t);
  for (let j = 0; j < 6; j++) ser += LANCZOS[j] / ++y;
  return L(2.5066282746310005 * ser / x) - t;
}
const lfactorial = (n) => (n < 0 ? NaN : lgamma(n + 1));
const lchoose = (n, k) => lfactorial(n) - lfactorial(k) - lfactorial(n - k);
const lbeta = (a, b) => lgamma(a) + lgamma(b) - lgamma(a + b);

// Every density returns its terms in the reference's left-to-right order (fp64 addition is not associative).
const ld = {
  lgamma, lfactorial, lchoose, lbeta,

  // ---- continuous
  norm: (x, m, s) => -0.5 * L(2 * PI) - L(s) - sq(x - m) / (2 * s * s),
  unif: (x, lo, hi) => ((x < lo || x > hi) ? NEG : L(1 / (hi - lo))),
  beta(x, a, b) {
    if (x > 1 || x < 0) return NEG;
    return (a === 1 && b === 1) ? 0 : (a - 1) * L(x) + (b - 1) * L(1 - x) - lbeta(a, b);
  },
  cauchy: (x, loc, s) => L(s) - L(sq(x - loc) + sq(s)) - L(PI),
  bivarnorm(x, m, s, rho) {
    const d0 = x[0] - m[0], d1 = x[1] - m[1];
    const z = sq(d0) / sq(s[0]) + sq(d1) / sq(s[1]) - (2 * rho * d0 * d1) / (s[0] * s[1]);
    const norm_const = -(L(2) + L(PI) + L(s[0]) + L(s[1]) + 0.5 * L(1 - sq(rho)));
    return norm_const - z / (2 * (1 - sq(rho)));
  },
  laplace: (x, loc, s) => (-Math.abs(x - loc) / s) - L(2 * s),
  gamma(x, k, rate) {
    const theta = 1 / rate;
    if (x < 0) return NEG;
    return (x === 0 && k === 1) ? -L(theta) : (k - 1) * L(x) - x / theta - lgamma(k) - k * L(theta);
  },
  invgamma: (x, k, s) => (x <= 0 ? NEG : -(k + 1) * L(x) - s / x - lgamma(k) + k * L(s)),
  lnorm: (x, m, s) => (x <= 0 ? NEG : -L(x) - 0.5 * L(2 * PI) - L(s) - sq(L(x) - m) / (2 * s * s)),
  pareto: (x, xm, k) => (x < xm ? NEG : L(k) + k * L(xm) - (k + 1) * L(x)),
  t(x, loc, s, nu) {
    if (nu > 1e100) nu = 1e100;
    return lgamma((nu + 1) / 2) - lgamma(nu / 2) - L(Math.sqrt(PI * nu) * s) + L(Math.pow(1 + (1 / nu) * sq((x - loc) / s), -(nu + 1) / 2));
  },
  weibull(x, k, lambda) {      // R's dweibull(log = TRUE)
    if (x < 0) return NEG;
    if (x === 0 && k < 1) return Infinity;
    const r = x / lambda, a = Math.pow(r, k - 1);
    return -(a * r) + L(k * a / lambda);
  },
  logis(x, loc, s) {           // R's dlogis(log = TRUE)
    const z = Math.abs((x - loc) / s), f = 1.0 + Math.exp(-z);
    return -(z + L(s * f * f));
  },
  dirichlet(x, alpha) {
    let a_sum = 0, lg_sum = 0, w_sum = 0;
    for (let i = 0; i < alpha.length; i++) {
      a_sum += alpha[i];
      lg_sum += lgamma(alpha[i]);
      w_sum += (alpha[i] - 1) * L(x[i]);
    }
    return lgamma(a_sum) - lg_sum + w_sum;
  },
  exp: (x, rate) => (x < 0 ? NEG : L(rate) - rate * x),

  // ---- discrete
  bern: (x, p) => (!(x === 0 || x === 1) ? NEG : L(x * p + (1 - x) * (1 - p))),
  cat: (x, p) => ((x < 1 || x > p.length) ? NEG : L(p[x - 1])),
  binom(x, n, p) {
    if (x > n || x < 0) return NEG;
    if (p === 0 || p === 1) return (n * p) === x ? 0 : NEG;
    return lchoose(n, x) + x * L(p) + (n - x) * L(1 - p);
  },
  nbinom: (x, r, p) => (x < 0 ? NEG : lchoose(x + r - 1, r - 1) + x * L(1 - p) + r * L(p)),
  hyper: (x, m, n, k) => ((x < 0 || x > k) ? NEG : lchoose(m, x) + lchoose(n, k - x) - lchoose(m + n, k)),
  pois: (x, lambda) => (x < 0 ? NEG : L(lambda) * x - lambda - lfactorial(x)),
};
ld.dexp = ld.laplace;
module.exports = ld;
