'use strict';
/*
 * ref_models.js -- TEST INFRASTRUCTURE ONLY.
 * The BASELINE.json models written the way a bayes.js user writes them
 * (README.md:18-43, 149-164): plain JS log_post(state, data) closures over the
 * reference's own `ld` object.  ref_harness.js feeds these to the UNMODIFIED
 * reference sampler; the registry ids are the ones include/amwg.h exposes.
 */
const DEFAULT_HYPER = { normal: [0, 100, 0, 100], beta_bern: [2, 2], hier_normal: [0, 100, 0, 100, 10], pois_glm: [0, 10] };

module.exports = function (ld, hyperOverride) {
  const H = (m) => (hyperOverride && hyperOverride[m]) || DEFAULT_HYPER[m];
  return {
    // cfg1/cfg2 -- README.md:22-36
    normal: {
      params: () => ({ mu: { type: 'real' }, sigma: { type: 'real', lower: 0 } }),
      log_post: function (s, d) {
        let lp = 0;
        const h = H('normal');
        lp += ld.norm(s.mu, h[0], h[1]);
        lp += ld.unif(s.sigma, h[2], h[3]);
        for (let i = 0; i < d.x.length; i++) lp += ld.norm(d.x[i], s.mu, s.sigma);
        return lp;
      },
    },
    // cfg3 -- README.md:149-164
    beta_bern: {
      params: () => ({ theta: { type: 'real', lower: 0, upper: 1 } }),
      log_post: function (s, d) {
        let lp = 0;
        const h = H('beta_bern');
        lp += ld.beta(s.theta, h[0], h[1]);
        const n = d.x.length;
        for (let i = 0; i < n; i++) lp += ld.bern(d.x[i], s.theta);
        return lp;
      },
    },
    // cfg4 -- SURVEY.md §8(d): theta[G] + mu + sigma  (G=32 -> 34 scalar components)
    hier_normal: {
      params: (d) => ({ theta: { type: 'real', dim: [d.G] }, mu: { type: 'real' }, sigma: { type: 'real', lower: 0, init: 1 } }),
      log_post: function (s, d) {
        let lp = 0;
        const h = H('hier_normal');
        lp += ld.norm(s.mu, h[0], h[1]);
        lp += ld.unif(s.sigma, h[2], h[3]);
        for (let k = 0; k < d.G; k++) lp += ld.norm(s.theta[k], s.mu, h[4]);
        for (let i = 0; i < d.y.length; i++) lp += ld.norm(d.y[i], s.theta[d.g[i]], s.sigma);
        return lp;
      },
    },
    // cfg5 -- SURVEY.md §8(d): beta[8] real + int change point
    pois_glm: {
      params: (d) => ({ beta: { type: 'real', dim: [8], init: 0 }, cp: { type: 'int', lower: 0, upper: d.y.length - 1 } }),
      log_post: function (s, d) {
        let lp = 0;
        const N = d.y.length, K = d.K;
        const h = H('pois_glm');
        for (let k = 0; k < 8; k++) lp += ld.norm(s.beta[k], h[0], h[1]);
        lp += ld.unif(s.cp, 0, N - 1);
        for (let i = 0; i < N; i++) {
          let eta = 0;
          for (let k = 0; k < K; k++) eta += d.X[i * K + k] * s.beta[k];
          if (i >= s.cp) eta += s.beta[7];
          lp += ld.pois(d.y[i], Math.exp(eta));
        }
        return lp;
      },
    },
  };
};
