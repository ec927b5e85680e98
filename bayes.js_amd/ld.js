'use strict';
/*
 * ld.js -- host-side (JavaScript) versions of the log densities that are on the GPU path, so
 * a user's log_post(state, data) closure written against the reference's `ld` object
 * (distributions.js) still runs on the host, e.g. to evaluate a density by hand.  The sampler
 * itself never calls these: it runs the HIP twins in csrc/amwg_ld.h.
 * Parameterisation and expression order follow distributions.js:63-92 (lgamma, lfactorial,
 * lbeta), :104-113 (beta), :119-121 (norm), :221-223 (unif), :228-230 (bern), :282-284 (pois).
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
const lbeta = (a, b) => lgamma(a) + lgamma(b) - lgamma(a + b);

const ld = {
  lgamma, lfactorial, lbeta,
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
module.exports = ld;
