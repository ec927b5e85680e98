// amwg_models.h -- the built-in log_post functors (the user's JS closure, mcmc.js:958-960,
// for the BASELINE.json model families).  Each model gives
//   prior(S, mc)            sequential sum of the prior terms, in the closure's order
//   begin(S, mc, d) -> Pass loop-invariant values of one pass over the data
//   term<FAST>(pass, i)     the i-th observation's log density
// and the generic log_post() in amwg_kernel.h adds them in the documented order.
// S(p) reads scalar component p of this chain's state.
#pragma once
#include "amwg_div.h"
#include "amwg_ld.h"
#include "amwg_types.h"

namespace amwg {

// LDS-resident view of this chain's state: component p at S.base[p].  Chains are laid out
// [chain][stride] with an ODD stride (in doubles): lanes that own different chains and read the
// same component hit 32 distinct 8-byte bank slots, and the G lanes of one chain that gather
// different components (theta[g_i]) read consecutive addresses -- conflict-free both ways.
struct StateView {
  const double *base;
  __device__ __forceinline__ double operator()(int p) const { return base[p]; }
};

// ld.norm(v, 0|m, sd) with constant sd (a prior): c_sd - (v-m)^2 / (2*sd*sd)
__device__ __forceinline__ double norm_const_sd(double v, double m, double c_sd, double den) {
  const double t = v - m;
  return c_sd - (t * t) / den;
}

// ---------------------------------------------------------------------------------------------
// x_i ~ norm(mu, sigma); mu ~ norm(m0,s0); sigma ~ unif(a,b)              README.md:22-36
struct NormalModel {
  static constexpr bool kDataInLds = true;
  struct Pass { double mu, c, den; Reciprocal y; bool fast; const double *x; };
  __host__ __device__ static size_t lds_bytes(int n_obs, int, int) { return (size_t)n_obs * 8; }
  __device__ static void stage(unsigned char *smem, const DataRef &d, int tid, int nt, int) {
    double *dst = reinterpret_cast<double *>(smem);
    for (int i = tid; i < d.n_obs; i += nt) dst[i] = d.x[i];
  }
  __device__ __forceinline__ static double prior(const StateView &S, const ModelConsts &mc, const DataRef &) {
    double lp = 0;
    lp += norm_const_sd(S(0), mc.m0, mc.c0, mc.den0);
    const double sigma = S(1);
    lp += (sigma < mc.ua || sigma > mc.ub) ? -kInf : mc.lunif;
    return lp;
  }
  __device__ __forceinline__ static Pass begin(const StateView &S, const ModelConsts &mc, const DataRef &,
                                               const unsigned char *smem) {
    Pass ps;
    ps.mu = S(0);
    const double sd = S(1);
    ps.c = norm_c(mc.neg_half_log_2pi, sd);
    ps.den = norm_den(sd);
    ps.y = make_reciprocal(ps.den);
    ps.fast = !mc.exact_division && mc.data_mid_range && mid_range(ps.den) &&
              (ps.mu == 0 || mid_range(__builtin_fabs(ps.mu)));
    ps.x = reinterpret_cast<const double *>(smem);
    return ps;
  }
  template <bool FAST>
  __device__ __forceinline__ static double term(const Pass &ps, int i) {
    const double t = ps.x[i] - ps.mu;
    const double tt = t * t;
    return ps.c - (FAST ? div_by_invariant(tt, ps.den, ps.y) : tt / ps.den);
  }
};

// ---------------------------------------------------------------------------------------------
// x_i ~ bern(theta); theta ~ beta(a,b)                                    README.md:149-164
// ld.bern(x,p) = log(x*p + (1-x)*(1-p)) is exactly log(p) for x=1 and log(1-p) for x=0
// (1*p + 0*(1-p) = p + 0 = p), so the two logs are hoisted and selected per observation.
struct BetaBernModel {
  static constexpr bool kDataInLds = true;
  struct Pass { double l1, l0; const uint8_t *x; const uint32_t *bits; bool has_invalid; };
  // one lane per chain reads the observations as bits through the scalar cache: no LDS copy
  __host__ __device__ static size_t lds_bytes(int n_obs, int, int lanes) { return lanes == 1 ? 0 : (((size_t)n_obs + 15) & ~(size_t)15); }
  __device__ static void stage(unsigned char *smem, const DataRef &d, int tid, int nt, int lanes) {
    if (lanes == 1) return;
    for (int i = tid; i < d.n_obs; i += nt) smem[i] = d.xb[i];
  }
  __device__ __forceinline__ static double prior(const StateView &S, const ModelConsts &mc, const DataRef &) {
    const double th = S(0);
    double lp = 0;
    if (th > 1 || th < 0) lp += -kInf;
    else if (mc.ba == 1 && mc.bb == 1) lp += 0.0;
    else lp += (mc.ba - 1) * log_v8(th) + (mc.bb - 1) * log_v8(1 - th) - mc.lbeta_ab;
    return lp;
  }
  __device__ __forceinline__ static Pass begin(const StateView &S, const ModelConsts &mc, const DataRef &d,
                                               const unsigned char *smem) {
    Pass ps;
    const double th = S(0);
    ps.l1 = log_v8(th);       // x = 1: log(1*th + 0*(1-th))
    ps.l0 = log_v8(1 - th);   // x = 0: log(0*th + 1*(1-th))
    ps.x = smem;
    ps.bits = d.xw;
    ps.has_invalid = mc.has_invalid != 0;
    return ps;
  }
  template <bool FAST>
  __device__ __forceinline__ static double term(const Pass &ps, int i) { return ps.x[i] ? ps.l1 : ps.l0; }

  // Sequential sum for ONE lane per chain.  Every lane of the wave adds the same observation at
  // the same time, so the observation bit is wave-uniform: it is read through the scalar path
  // (32 observations per s_load'ed word) and selects, with a scalar branch, WHICH per-lane
  // register (log theta or log(1-theta)) the single v_add_f64 of that observation adds --
  // 1 vector instruction per observation instead of compare + 2 selects + add.  The adds are
  // inline asm so the compiler cannot turn the uniform branch back into per-lane selects.
  // acc += bit b of the wave-uniform word w ? l1 : l0, as ONE vector instruction: the scalar unit
  // tests the bit and branches around the other add.  One asm statement per observation so both
  // arms write the same register (no phi copies) and the compiler cannot if-convert the branch.
#define AMWG_BERN_ADD(B)                                                                              \
  asm volatile("s_bitcmp1_b32 %3, " #B "\n\ts_cbranch_scc1 1f\n\tv_add_f64 %0, %0, %2\n\ts_branch 2f\n"  \
               "1:\n\tv_add_f64 %0, %0, %1\n2:"                                                        \
               : "+v"(acc) : "v"(l1), "v"(l0), "s"(w) : "scc")
  __device__ __forceinline__ static double pass_one_lane(const Pass &ps, int n_obs, double acc) {
    const double l1 = ps.l1, l0 = ps.l0;
    const int nw = n_obs >> 5;
    uint32_t w_next = nw > 0 ? ps.bits[0] : 0u;
    for (int k = 0; k < nw; ++k) {
      const uint32_t w = __builtin_amdgcn_readfirstlane(w_next);
      if (k + 1 < nw || (n_obs & 31)) w_next = ps.bits[k + 1];
      AMWG_BERN_ADD(0);  AMWG_BERN_ADD(1);  AMWG_BERN_ADD(2);  AMWG_BERN_ADD(3);
      AMWG_BERN_ADD(4);  AMWG_BERN_ADD(5);  AMWG_BERN_ADD(6);  AMWG_BERN_ADD(7);
      AMWG_BERN_ADD(8);  AMWG_BERN_ADD(9);  AMWG_BERN_ADD(10); AMWG_BERN_ADD(11);
      AMWG_BERN_ADD(12); AMWG_BERN_ADD(13); AMWG_BERN_ADD(14); AMWG_BERN_ADD(15);
      AMWG_BERN_ADD(16); AMWG_BERN_ADD(17); AMWG_BERN_ADD(18); AMWG_BERN_ADD(19);
      AMWG_BERN_ADD(20); AMWG_BERN_ADD(21); AMWG_BERN_ADD(22); AMWG_BERN_ADD(23);
      AMWG_BERN_ADD(24); AMWG_BERN_ADD(25); AMWG_BERN_ADD(26); AMWG_BERN_ADD(27);
      AMWG_BERN_ADD(28); AMWG_BERN_ADD(29); AMWG_BERN_ADD(30); AMWG_BERN_ADD(31);
    }
    uint32_t w = __builtin_amdgcn_readfirstlane(w_next);
    for (int b = 0; b < (n_obs & 31); ++b) {
      AMWG_BERN_ADD(0);
      w >>= 1;
    }
    return acc;
  }
#undef AMWG_BERN_ADD
};

// ---------------------------------------------------------------------------------------------
// y_i ~ norm(theta[g_i], sigma); theta_g ~ norm(mu,10); mu ~ norm(0,100); sigma ~ unif(0,100)
// components: theta[0..G-1], mu, sigma                                    SURVEY.md §8(d) cfg4
struct HierNormalModel {
  static constexpr bool kDataInLds = true;
  struct Pass { double c, den; Reciprocal y; bool fast; const double *x; const uint8_t *g; StateView S; };
  __host__ __device__ static size_t lds_bytes(int n_obs, int, int) { return (size_t)n_obs * 8 + (((size_t)n_obs + 15) & ~(size_t)15); }
  __device__ static void stage(unsigned char *smem, const DataRef &d, int tid, int nt, int) {
    double *dst = reinterpret_cast<double *>(smem);
    uint8_t *gd = smem + (size_t)d.n_obs * 8;
    for (int i = tid; i < d.n_obs; i += nt) { dst[i] = d.x[i]; gd[i] = d.xb[i]; }
  }
  __device__ __forceinline__ static double prior(const StateView &S, const ModelConsts &mc, const DataRef &d) {
    const double mu = S(d.G), sigma = S(d.G + 1);
    double lp = 0;
    lp += norm_const_sd(mu, mc.m0, mc.c0, mc.den0);
    lp += (sigma < mc.ua || sigma > mc.ub) ? -kInf : mc.lunif;
    for (int k = 0; k < d.G; ++k) lp += norm_const_sd(S(k), mu, mc.c1, mc.den1);
    return lp;
  }
  __device__ __forceinline__ static Pass begin(const StateView &S, const ModelConsts &mc, const DataRef &d,
                                               const unsigned char *smem) {
    Pass ps;
    const double sd = S(d.G + 1);
    ps.c = norm_c(mc.neg_half_log_2pi, sd);
    ps.den = norm_den(sd);
    ps.y = make_reciprocal(ps.den);
    bool ok = !mc.exact_division && mc.data_mid_range && mid_range(ps.den);
    for (int k = 0; k < d.G; ++k) { const double th = S(k); ok = ok && (th == 0 || mid_range(__builtin_fabs(th))); }
    ps.fast = ok;
    ps.x = reinterpret_cast<const double *>(smem);
    ps.g = smem + (size_t)d.n_obs * 8;
    ps.S = S;
    return ps;
  }
  template <bool FAST>
  __device__ __forceinline__ static double term(const Pass &ps, int i) {
    const double t = ps.x[i] - ps.S(ps.g[i]);
    const double tt = t * t;
    return ps.c - (FAST ? div_by_invariant(tt, ps.den, ps.y) : tt / ps.den);
  }
};

// ---------------------------------------------------------------------------------------------
// y_i ~ pois(exp(sum_k X[i][k]*beta[k] + [i >= cp]*beta[7])); beta_k ~ norm(0,10); cp ~ unif(0,N-1)
// components: beta[0..7], cp (int)                                        SURVEY.md §8(d) cfg5
// The design matrix (3.6 MB at N=5e4) is read straight from L2/MALL, stored column-major so the
// G lanes of a chain read consecutive observations of one column (coalesced); the per-observation
// exp+log dominate the arithmetic by two orders of magnitude.
struct PoisGlmModel {
  static constexpr bool kDataInLds = false;
  struct Pass { double b[8]; double cp; const double *X, *y, *lfact; int N; };
  __host__ __device__ static size_t lds_bytes(int, int, int) { return 0; }
  __device__ static void stage(unsigned char *, const DataRef &, int, int, int) {}
  __device__ __forceinline__ static double prior(const StateView &S, const ModelConsts &mc, const DataRef &) {
    double lp = 0;
    for (int k = 0; k < 8; ++k) lp += norm_const_sd(S(k), mc.m0, mc.c0, mc.den0);
    const double cp = S(8);
    lp += (cp < 0 || cp > mc.cp_upper) ? -kInf : mc.lunif_cp;
    return lp;
  }
  __device__ __forceinline__ static Pass begin(const StateView &S, const ModelConsts &, const DataRef &d,
                                               const unsigned char *) {
    Pass ps;
#pragma unroll
    for (int k = 0; k < 8; ++k) ps.b[k] = S(k);
    ps.cp = S(8);
    ps.X = d.x; ps.y = d.y; ps.lfact = d.lfact; ps.N = d.n_obs;
    return ps;
  }
  template <bool FAST>
  __device__ __forceinline__ static double term(const Pass &ps, int i) {
    double eta = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) eta += ps.X[(size_t)k * ps.N + i] * ps.b[k];
    if ((double)i >= ps.cp) eta += ps.b[7];
    const double lam = exp_v8(eta);
    return log_v8(lam) * ps.y[i] - lam - ps.lfact[i];
  }
};

}  // namespace amwg
