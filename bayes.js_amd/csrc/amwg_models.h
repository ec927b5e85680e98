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
#include "amwg_pass.h"
#include "amwg_twoval.h"
#include "amwg_types.h"
#include "amwg_window.h"

namespace amwg {

// ld.norm(v, 0|m, sd) with constant sd (a prior): c_sd - (v-m)^2 / (2*sd*sd).  The divisor is a hyper-parameter, so its
// double-double reciprocal comes from the host (ModelConsts) and the quotient is the 4-operation correctly rounded one of
// amwg_div.h whenever divisor and numerator are inside its range -- the same bits as '/', 13 instructions less per prior term.
__device__ __forceinline__ double norm_const_sd(double v, double m, double c_sd, double den, double y_hi, double y_lo, int den_ok) {
  const double t = v - m;
  const double tt = t * t;
  const double q = (den_ok && wide_range(tt)) ? div_by_invariant(tt, den, Reciprocal{y_hi, y_lo}) : tt / den;
  return c_sd - q;
}

// The sd-dependent invariants of a pass -- c = -0.5*log(2*pi) - log(sd), den = 2*sd*sd, 1/den as a double-double -- kept per
// chain across evaluations: most updates of a hierarchical model propose another component, sd is unchanged and so are these
// (pure functions of sd; ~80 instructions incl. V8's log and an IEEE division).  sd starts as NaN, which never compares equal.
struct NormCache {
  double sd, c, den;
  Reciprocal y;
  bool den_ok;
};
__device__ __forceinline__ NormCache norm_cache_init() { return NormCache{__builtin_nan(""), 0.0, 0.0, Reciprocal{0.0, 0.0}, false}; }
AMWG_HD NormCache norm_cache_make(double sd, double neg_half_log_2pi) {
  NormCache k;
  k.sd = sd;
  k.c = norm_c(neg_half_log_2pi, sd);
  k.den = norm_den(sd);
  k.y = make_reciprocal(k.den);
  k.den_ok = mid_range(k.den);
  return k;
}
// out of line for models whose sd changes rarely (one update in 34 of the hierarchical model): V8's log and an IEEE division, with their
// code and constants, stay out of the hot loop.  The values come back in registers (two calls: a seven-member struct would travel through
// the stack).
struct NormCacheA { double c, den; };
AMWG_HD_OUTLINE NormCacheA norm_cache_cold_a(double sd, double neg_half_log_2pi) { return NormCacheA{norm_c(neg_half_log_2pi, sd), norm_den(sd)}; }
AMWG_HD_OUTLINE Reciprocal norm_cache_cold_b(double den) { return make_reciprocal(den); }
template <bool COLD = false>
__device__ __forceinline__ void norm_cache_update(NormCache &k, double sd, double neg_half_log_2pi) {
  if (sd != k.sd) {
    if constexpr (COLD) {
      const NormCacheA a = norm_cache_cold_a(sd, neg_half_log_2pi);
      k.sd = sd; k.c = a.c; k.den = a.den;
      k.y = norm_cache_cold_b(a.den);
      k.den_ok = mid_range(a.den);
    } else {
      k = norm_cache_make(sd, neg_half_log_2pi);
    }
  }
}


// (norm_sq_pass_wave -- the certified pass of the Normal family for the 64 chains of a wavefront at once -- lives in amwg_pass.h: translated closures that end in the
// same likelihood loop run it too, amwg_user.h norm_tail_approx)

// ---------------------------------------------------------------------------------------------
// x_i ~ norm(mu, sigma); mu ~ norm(m0,s0); sigma ~ unif(a,b)              README.md:22-36
struct NormalModel {
  static constexpr bool kSplitPrior = false;
  static constexpr bool kUser = false, kHasFast = true, kOneLanePass = false;
  static constexpr int kDerived = 0;
  static constexpr bool kHasBinary = false;   // real / int parameters only: the BinaryStepper branch is not compiled in
  static constexpr int kMaxThreads = 1024;   // workgroup size cap (instantiated per size class 256 / 512 / 1024, amwg_kernels.hip)
  static constexpr int kUnroll = 8;   // independent terms in flight per lane (ILP across the division chains)
  struct Pass { double mu, c, den; Reciprocal y; bool fast; const double *x; };
  // one lane per chain: the EXPRESSION's pass reads the observations through the scalar cache (norm_pass_uniform); the certified pass (norm_sq_pass_wave) reads
  // a tile in LDS when the data fits beside the stepper state of a full workgroup (else the array in global memory)
  // (only that pass reads the one-lane tile: the host asks for it -- DataRef::pad = 1 -- when the launch is of the certified kernel in workgroups of at most 256 threads;
  // the expression-in-every-update kernels, exact_division and the larger classes neither stage nor reserve it -- round-5 advisor finding)
  static constexpr size_t kOneLaneTileLimit = 96 * 1024;
  __host__ __device__ static size_t one_lane_tile_bytes(int n_obs) { return (size_t)n_obs * 8 <= kOneLaneTileLimit ? (size_t)n_obs * 8 : 0; }
  __host__ __device__ static size_t lds_bytes(int n_obs, int, int lanes) { return lanes == 1 ? 0 : (size_t)n_obs * 8; }
  static constexpr bool kDynamicLds = true;
  __host__ __device__ static size_t lds_bytes_of(const DataRef &d, int lanes, int) { return lanes == 1 ? (d.pad > 0 ? one_lane_tile_bytes(d.n_obs) : 0) : (size_t)d.n_obs * 8; }
  __device__ static void stage(unsigned char *smem, const DataRef &d, int tid, int nt, int lanes) {
    if (lanes == 1 && lds_bytes_of(d, 1, nt) == 0) return;
    double *dst = reinterpret_cast<double *>(smem);
    for (int i = tid; i < d.n_obs; i += nt) dst[i] = d.x[i];
  }
  template <class C>
  __device__ __forceinline__ static double prior(const StateView &, const ModelConsts &mc, const DataRef &, C &kc) {
    double lp = 0;
    lp += norm_const_sd(kc.mu, mc.m0, mc.c0, mc.den0, mc.y0_hi, mc.y0_lo, mc.den0_ok);
    const double sigma = kc.sigma;
    lp += (sigma < mc.ua || sigma > mc.ub) ? -kInf : mc.lunif;
    return lp;
  }
  // the chain's two values are mirrored in registers (kept current by on_set): an evaluation does not wait for LDS to hand back what the
  // stepper has just stored
  static constexpr bool kTracksState = true;
  struct Cache { NormCache n; double mu, sigma; bool loaded; };
  __device__ __forceinline__ static Cache cache_init() { return Cache{norm_cache_init(), 0.0, 0.0, false}; }
  __device__ __forceinline__ static void on_set(Cache &k, int comp, double v, int, const DataRef &) { k.mu = comp == 0 ? v : k.mu; k.sigma = comp == 0 ? k.sigma : v; }
  static constexpr bool kMirrorCheck = true;
  template <int GL>
  __device__ __forceinline__ static bool mirror_ok(const Cache &k, const StateView &S, const DataRef &, int) {
    return !k.loaded || (f64_bits(k.mu) == f64_bits(S(0)) && f64_bits(k.sigma) == f64_bits(S(1)));
  }
  template <int GL>
  __device__ __forceinline__ static void load(Cache &k, const StateView &S, const ModelConsts &, const DataRef &, const unsigned char *, int) {
    if (!k.loaded) { k.mu = S(0); k.sigma = S(1); k.loaded = true; }
  }
  template <int GL>
  __device__ __forceinline__ static Pass begin(const StateView &, const ModelConsts &mc, const DataRef &d,
                                               const unsigned char *smem, Cache &kc) {
    Pass ps;
    NormCache &k = kc.n;
    ps.mu = kc.mu;
    norm_cache_update(k, kc.sigma, mc.neg_half_log_2pi);
    ps.c = k.c;
    ps.den = k.den;
    ps.y = k.y;
    ps.fast = !mc.exact_division && mc.data_mid_range && k.den_ok &&
              (ps.mu == 0 || mid_range(__builtin_fabs(ps.mu)));
    ps.x = GL == 1 ? d.x : reinterpret_cast<const double *>(smem);
    return ps;
  }
  template <bool FAST>
  __device__ __forceinline__ static double term(const Pass &ps, int i) {
    const double t = ps.x[i] - ps.mu;
    const double tt = t * t;
    return ps.c - (FAST ? div_by_invariant(tt, ps.den, ps.y) : tt / ps.den);
  }
  // the term-by-term pass with IEEE division, for evaluations whose values fall outside the range the 4-operation quotient needs: out of
  // line (scalar arguments, so that nothing travels through the stack), its code and registers are not part of the hot loop
  template <int G>
  __device__ inline __attribute__((noinline)) static double pass_slow_impl(const double *x, double mu, double c, double den, int n_obs, int sub, double acc) {
    for (int i = sub; i < n_obs; i += G) { const double t = x[i] - mu; acc += c - (t * t) / den; }
    return acc;
  }
  template <int G>
  __device__ __forceinline__ static double pass_slow(const Pass &ps, int n_obs, int sub, double acc) { return pass_slow_impl<G>(ps.x, ps.mu, ps.c, ps.den, n_obs, sub, acc); }
  // the fast pass, hand-pipelined (same operations and order as term<true> summed by pass_over_data)
  static constexpr bool kStagedFast = true;
  template <int G, int U = 8>
  __device__ __forceinline__ static double pass_fast(const Pass &ps, int n_obs, int sub, double acc) {
    if constexpr (G == 1) return norm_pass_uniform<8>(ps.x, ps.mu, ps.c, ps.den, ps.y, n_obs, acc);   // ps.x = the global array
    else return norm_pass_staged<G, U, false>(ps.x, nullptr, StateView{nullptr}, ps.mu, ps.c, ps.den, ps.y, n_obs, sub, acc);
  }
  // CERTIFIED DECISIONS (one lane per chain; amwg_kernel.h).  log_post of the state the stepper has just stored, as  prior + n c - S2 / den  with
  // S2 = sum (x_i - mu)^2 (amwg_pass.h norm_sq_pass_uniform: two operations per observation), together with a bound eps on how far BOTH this value and
  // the one the reference's expression gives -- prior + term_0 + term_1 + ... in order, term_i = c - RN((x_i - mu)^2 / den) (log_post above) -- can
  // lie from each other.  With u = 2^-53, Q = S2 / den, mag = |prior| + n |c| + Q:
  //     term by term:  n correctly rounded quotients (u Q), n subtractions (u (n |c| + Q)), n + 1 additions whose partial sums stay below mag (n u mag)
  //     here:          eight partial sums of n / 8 non-negative terms each ((n / 8 + 3) u S2), the squares taken exactly inside the fma where the other path
  //                    rounds them (u S2), 1 / den correctly rounded and one product (2 u Q), n c and two additions (3 u mag)
  // i.e. |difference| <= (1.125 n + 16) u mag; eps = (2 n + 64) u (|prior| + n |c| + 2 Q) 1.25 leaves a factor of two.  A non-finite value anywhere
  // makes eps non-finite, which the stepper reads as "evaluate the expression".
  static constexpr bool kCertified = true;
  static constexpr int kCertifiedLanes = 1;
  struct Approx { double value, eps; };
  template <int G, int BT>
  __device__ __forceinline__ static Approx log_post_approx(Cache &kc, const StateView &S, const ModelConsts &mc, const DataRef &d, const unsigned char *smem, int sub) {
    static_assert(G == 1, "the certified pass of the Normal family is the one-lane one");
    load<G>(kc, S, mc, d, smem, sub);
    norm_cache_update(kc.n, kc.sigma, mc.neg_half_log_2pi);
    const double P = prior(S, mc, d, kc);
    // (every lane of the wavefront takes part: the caller has made sure of that.  The wavefront's pass keeps 64 partial sums per lane: workgroups of up to 256
    // threads, whose lanes have 512 registers; the larger classes read the observations one at a time through the scalar path)
    double S2;
    if (mc.sufficient) {
      // amwg_options::sufficient_statistics (opt-in, a third tier): S2 = SS + n (xbar - mu)^2 from the two sufficient statistics of the data, no pass.  As a real number
      // this is sum (x_i - mu)^2; the expression squares the ROUNDED differences RN(x_i - mu), which moves its real-number target by at most 2 u S2; the three
      // operations here add 3 u S2 and the host's quad-precision xbar and SS 2^-100: inside the (n / 8 + 9) u mag the derivation above grants this side
      const double dm = (mc.suff_xbar_hi - kc.mu) + mc.suff_xbar_lo;
      S2 = __builtin_fma((double)d.n_obs, dm * dm, mc.suff_ss);
    } else
    // (the wavefront's pass in the 256- and 512-thread classes: blocks of 16 observations per lane where a lane has 512 registers, of 8 where it has 256)
    if constexpr (BT <= 512) S2 = norm_sq_pass_wave<(BT <= 256 ? kWaveBlock : 8)>(lds_bytes_of(d, 1, 0) ? reinterpret_cast<const double *>(smem) : d.x, kc.mu, d.n_obs, wave_scratch_of(d));
    else S2 = norm_sq_pass_uniform<8>(d.x, kc.mu, d.n_obs);
    const double n = (double)d.n_obs;
    const double Q = S2 * kc.n.y.hi, nc = n * kc.n.c;
    const double mag = __builtin_fabs(P) + __builtin_fabs(nc) + 2.0 * Q;
    return Approx{(P + nc) - Q, (2.0 * n + 64.0) * 0x1p-53 * mag * 1.25};
  }
};

// ---------------------------------------------------------------------------------------------
// x_i ~ bern(theta); theta ~ beta(a,b)                                    README.md:149-164
// ld.bern(x,p) = log(x*p + (1-x)*(1-p)) is exactly log(p) for x=1 and log(1-p) for x=0
// (1*p + 0*(1-p) = p + 0 = p), so the two logs are hoisted and selected per observation.
struct BetaBernModel {
  static constexpr bool kSplitPrior = false;
  static constexpr bool kUser = false, kHasFast = false, kOneLanePass = true;
  static constexpr int kDerived = 0;
  static constexpr bool kHasBinary = false;   // real / int parameters only: the BinaryStepper branch is not compiled in
  static constexpr int kMaxThreads = 1024;   // workgroup size cap (instantiated per size class 256 / 512 / 1024, amwg_kernels.hip)
  static constexpr int kUnroll = 8;
  struct Pass { double l1, l0; const uint8_t *x; const uint32_t *bits; bool has_invalid, fast_forward; BitData B; };
  // one lane per chain: the observations as bits plus their prefix popcounts in LDS (two_valued_sum above); the
  // term-by-term pass (exact_division = 1) reads the bits through the scalar cache instead
  __host__ __device__ static size_t words(int n_obs) { return (size_t)n_obs / 32 + 2; }
  static constexpr size_t kFfLdsLimit = 120 * 1024;   // beyond this the six bit arrays stay in HBM/L2 (read through L1)
  __host__ __device__ static size_t lds_bytes(int n_obs, int, int lanes) {
    if (lanes == 1) { const size_t b = 6 * words(n_obs) * 4; return b <= kFfLdsLimit ? ((b + 15) & ~(size_t)15) : 0; }
    return ((size_t)n_obs + 15) & ~(size_t)15;
  }
  __device__ static void stage(unsigned char *smem, const DataRef &d, int tid, int nt, int lanes) {
    if (lanes == 1) {   // d.arr[0]: the six arrays of BitData back to back (w, pre, om1, po1, om0, po0)
      if (lds_bytes(d.n_obs, 0, 1) == 0) return;
      uint32_t *dst = reinterpret_cast<uint32_t *>(smem);
      const uint32_t *src = static_cast<const uint32_t *>(d.arr[0]);
      for (int k = tid; k < (int)(6 * words(d.n_obs)); k += nt) dst[k] = src[k];
      return;
    }
    for (int i = tid; i < d.n_obs; i += nt) smem[i] = d.xb[i];
  }
  template <class C>
  __device__ __forceinline__ static double prior(const StateView &S, const ModelConsts &mc, const DataRef &, C &) {
    const double th = S(0);
    double lp = 0;
    if (th > 1 || th < 0) lp += -kInf;
    else if (mc.ba == 1 && mc.bb == 1) lp += 0.0;
    else lp += (mc.ba - 1) * log_v8(th) + (mc.bb - 1) * log_v8(1 - th) - mc.lbeta_ab;
    return lp;
  }
  struct Cache {};
  __device__ __forceinline__ static Cache cache_init() { return Cache{}; }
  template <int GL>
  __device__ __forceinline__ static Pass begin(const StateView &S, const ModelConsts &mc, const DataRef &d,
                                               const unsigned char *smem, Cache &) {
    Pass ps;
    const double th = S(0);
    ps.l1 = log_v8(th);       // x = 1: log(1*th + 0*(1-th))
    ps.l0 = log_v8(1 - th);   // x = 0: log(0*th + 1*(1-th))
    ps.x = smem;
    ps.bits = d.xw;
    ps.has_invalid = mc.has_invalid != 0;
    ps.fast_forward = !mc.exact_division;
    {
      const size_t W = words(d.n_obs);
      const uint32_t *base = lds_bytes(d.n_obs, 0, 1) ? reinterpret_cast<const uint32_t *>(smem) : static_cast<const uint32_t *>(d.arr[0]);
      ps.B.w = base; ps.B.pre = base + W; ps.B.om1 = base + 2 * W; ps.B.po1 = base + 3 * W; ps.B.om0 = base + 4 * W; ps.B.po0 = base + 5 * W;
      ps.B.n = d.n_obs;
    }
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
  // Sequential sum for ONE lane per chain (the reference's order).  Every lane of the wave adds
  // the same observation at the same time, so the observation bit is wave-uniform: it travels
  // through the scalar unit and only decides WHICH per-lane register (log theta or
  // log(1-theta)) the observation's single v_add_f64 adds -- 1 vector instruction per
  // observation instead of compare + 2 selects + add.  On CDNA a SIMD issues at most one scalar
  // instruction per 4-cycle turn, the same cadence as one fp64 add, so a per-observation
  // test-and-branch (>= 3 scalar/branch issues) runs at 12 cycles per observation (measured).
  // Instead 8 observations are dispatched at once: the next data byte indexes a table of 256
  // straight-line blocks (8 adds + branch back, 68 bytes each) via s_setpc_b64 -- ~0.9 scalar
  // issues per observation, leaving the fp64 adds as the bound.  Hand-written because the
  // compiler would turn the uniform choice back into per-lane selects.
#define AMWG_A0 "v_add_f64 %[acc], %[acc], %[l0]\n"
#define AMWG_A1 "v_add_f64 %[acc], %[acc], %[l1]\n"
#define AMWG_N0 AMWG_A0 AMWG_A0 AMWG_A0 AMWG_A0
#define AMWG_N1 AMWG_A1 AMWG_A0 AMWG_A0 AMWG_A0
#define AMWG_N2 AMWG_A0 AMWG_A1 AMWG_A0 AMWG_A0
#define AMWG_N3 AMWG_A1 AMWG_A1 AMWG_A0 AMWG_A0
#define AMWG_N4 AMWG_A0 AMWG_A0 AMWG_A1 AMWG_A0
#define AMWG_N5 AMWG_A1 AMWG_A0 AMWG_A1 AMWG_A0
#define AMWG_N6 AMWG_A0 AMWG_A1 AMWG_A1 AMWG_A0
#define AMWG_N7 AMWG_A1 AMWG_A1 AMWG_A1 AMWG_A0
#define AMWG_N8 AMWG_A0 AMWG_A0 AMWG_A0 AMWG_A1
#define AMWG_N9 AMWG_A1 AMWG_A0 AMWG_A0 AMWG_A1
#define AMWG_N10 AMWG_A0 AMWG_A1 AMWG_A0 AMWG_A1
#define AMWG_N11 AMWG_A1 AMWG_A1 AMWG_A0 AMWG_A1
#define AMWG_N12 AMWG_A0 AMWG_A0 AMWG_A1 AMWG_A1
#define AMWG_N13 AMWG_A1 AMWG_A0 AMWG_A1 AMWG_A1
#define AMWG_N14 AMWG_A0 AMWG_A1 AMWG_A1 AMWG_A1
#define AMWG_N15 AMWG_A1 AMWG_A1 AMWG_A1 AMWG_A1
#define AMWG_B(HI, LO) AMWG_N##LO AMWG_N##HI "s_branch 25b\n"     /* byte HI*16+LO: low nibble = first 4 observations */
#define AMWG_ROW(HI)                                                                                   \
  AMWG_B(HI, 0) AMWG_B(HI, 1) AMWG_B(HI, 2) AMWG_B(HI, 3) AMWG_B(HI, 4) AMWG_B(HI, 5) AMWG_B(HI, 6) AMWG_B(HI, 7) \
  AMWG_B(HI, 8) AMWG_B(HI, 9) AMWG_B(HI, 10) AMWG_B(HI, 11) AMWG_B(HI, 12) AMWG_B(HI, 13) AMWG_B(HI, 14) AMWG_B(HI, 15)
#define AMWG_BERN_ADD_BIT0                                                                             \
  asm volatile("s_bitcmp1_b32 %3, 0\n\ts_cbranch_scc1 1f\n\tv_add_f64 %0, %0, %2\n\ts_branch 2f\n"        \
               "1:\n\tv_add_f64 %0, %0, %1\n2:"                                                        \
               : "+v"(acc) : "v"(l1), "v"(l0), "s"(w) : "scc")
  __device__ __forceinline__ static double pass_one_lane(const Pass &ps, int n_obs, double acc) {
    if (ps.fast_forward) return two_valued_sum(acc, ps.l1, ps.l0, ps.B);
    return pass_one_lane_sequential(ps, n_obs, acc);
  }
  __device__ __attribute__((noinline)) static double pass_one_lane_sequential(const Pass &ps, int n_obs, double acc) {
    const double l1 = ps.l1, l0 = ps.l0;
    // wave-uniform by construction; make that explicit so they live in SGPRs
    const uint64_t pbits = (uint64_t)reinterpret_cast<uintptr_t>(ps.bits);
    const uint32_t plo = __builtin_amdgcn_readfirstlane((uint32_t)pbits), phi = __builtin_amdgcn_readfirstlane((uint32_t)(pbits >> 32));
    const uint32_t *bits = reinterpret_cast<const uint32_t *>((uintptr_t)(((uint64_t)phi << 32) | plo));
    n_obs = __builtin_amdgcn_readfirstlane(n_obs);
    const int nw = n_obs >> 5;      // full 32-observation words go through the byte-dispatch loop
    if (nw > 0) {
      asm volatile(
          "s_mov_b64 s[40:41], %[ptr]\n"
          "s_mov_b32 s42, %[nw]\n"
          "s_getpc_b64 s[44:45]\n"
          "10:\n"
          "s_add_u32 s44, s44, 30f-10b\n"      // s[44:45] = address of the block table
          "s_addc_u32 s45, s45, 0\n"
          "20:\n"                              // ---- next word: s[46:47] = {word, sentinel 1}
          "s_load_dword s46, s[40:41], 0x0\n"
          "s_mov_b32 s47, 1\n"
          "s_add_u32 s40, s40, 4\n"
          "s_addc_u32 s41, s41, 0\n"
          "s_waitcnt lgkmcnt(0)\n"
          "22:\n"                              // ---- next byte of the word
          "s_and_b32 s48, s46, 0xff\n"
          "s_mul_i32 s48, s48, 68\n"           // 8 * 8-byte VOP3 adds + 4-byte s_branch
          "s_add_u32 s48, s44, s48\n"
          "s_addc_u32 s49, s45, 0\n"
          "s_setpc_b64 s[48:49]\n"
          "25:\n"                              // ---- blocks return here
          "s_lshr_b64 s[46:47], s[46:47], 8\n"
          "s_cmp_lg_u64 s[46:47], 1\n"         // only the sentinel left => word done
          "s_cbranch_scc1 22b\n"
          "s_sub_u32 s42, s42, 1\n"
          "s_cmp_lg_u32 s42, 0\n"
          "s_cbranch_scc1 20b\n"
          "s_branch 40f\n"
          "30:\n"
          AMWG_ROW(0) AMWG_ROW(1) AMWG_ROW(2) AMWG_ROW(3) AMWG_ROW(4) AMWG_ROW(5) AMWG_ROW(6) AMWG_ROW(7)
          AMWG_ROW(8) AMWG_ROW(9) AMWG_ROW(10) AMWG_ROW(11) AMWG_ROW(12) AMWG_ROW(13) AMWG_ROW(14) AMWG_ROW(15)
          "40:\n"
          : [acc] "+v"(acc)
          : [l1] "v"(l1), [l0] "v"(l0), [ptr] "s"(bits), [nw] "s"(nw)
          : "s40", "s41", "s42", "s44", "s45", "s46", "s47", "s48", "s49", "scc", "memory");
    }
    if (n_obs & 31) {               // ragged tail, one test-and-branch per observation
      uint32_t w = __builtin_amdgcn_readfirstlane(bits[nw]);
      for (int b = 0; b < (n_obs & 31); ++b) {
        AMWG_BERN_ADD_BIT0;
        w >>= 1;
      }
    }
    return acc;
  }
#undef AMWG_BERN_ADD_BIT0
#undef AMWG_ROW
#undef AMWG_B
};

// ---------------------------------------------------------------------------------------------
// y_i ~ norm(theta[g_i], sigma); theta_g ~ norm(mu,10); mu ~ norm(0,100); sigma ~ unif(0,100)
// components: theta[0..G-1], mu, sigma                                    SURVEY.md §8(d) cfg4
struct HierNormalModel {
  static constexpr bool kUser = false, kHasFast = true, kOneLanePass = false;
  static constexpr int kDerived = 0;
  static constexpr bool kHasBinary = false;   // real / int parameters only: the BinaryStepper branch is not compiled in
  static constexpr int kMaxThreads = 1024;   // workgroup size cap (instantiated per size class 256 / 512 / 1024, amwg_kernels.hip)
  static constexpr int kUnroll = 8;
  struct Pass { double c, den, th_pass; Reciprocal y; bool fast, lane_const, regs, rows; const double *x; const uint8_t *g; StateView S; };
  __host__ __device__ static size_t lds_bytes(int n_obs, int, int) { return (size_t)n_obs * 8 + (((size_t)n_obs + 15) & ~(size_t)15); }
  // ROW LAYOUT (DataRef::pad = row pitch Rp > 0; chosen by the host for a chain on one wavefront whose labels repeat with the lane stride, i.e.
  // a lane that meets ONE group): the observations of lane j -- j, j + 64, ... -- stand side by side, row j of a [64][Rp] tile, Rp odd (lane
  // stride 8 Rp bytes: the 32 lanes of a half-wave land in 32 different bank pairs, and the U observations of a block are one base address
  // with immediate offsets).  What it buys is the lane-local re-evaluation below: any lane can read any other lane's observations side by side.
  // LDS: tile | the first 64 labels | per wavefront kMaxLocal rows of terms.
  static constexpr int kMaxLocal = 4;      // lanes whose sums are re-formed cooperatively; more stale lanes: the ordinary pass
  __host__ __device__ static int row_pitch(int n_obs) { return ((n_obs + 63) / 64) | 1; }
  __host__ __device__ static int local_rows(int groups) { const int per_group = groups > 0 ? 64 / groups : 1; const int r = per_group < 2 ? 2 : per_group; return r > kMaxLocal ? kMaxLocal : r; }      // term rows per wavefront: the lanes of one group (in pairs)
  __host__ __device__ static int term_pitch(int pitch) { return (pitch + 16 + 1) & ~1; }      // a row of terms: 16 spare slots (the adder reads ahead), 16-byte aligned
  // (+ one 256-uniform window per wavefront: the sweep kernel's stream, amwg_window.h)
  __host__ __device__ static size_t rows_window_offset(int pitch, int waves, int groups) { return (size_t)64 * pitch * 8 + 64 + (size_t)waves * local_rows(groups) * term_pitch(pitch) * 8; }
  __host__ __device__ static size_t rows_lds_bytes(int pitch, int waves, int groups) { return rows_window_offset(pitch, waves, groups) + (size_t)waves * 256 * 8; }
  using SweepStream = WindowStream;
  __device__ __forceinline__ static size_t window_offset(const DataRef &d, int waves) { return rows_window_offset(d.pad, waves, d.G); }
  // which parameter vector the sweep prefetch is for (amwg_kernel.h kSweep): theta, the first parameter, one entry per group
  __device__ __forceinline__ static int sweep_base(const DataRef &) { return 0; }
  __device__ __forceinline__ static int sweep_len(const DataRef &d) { return d.G; }
  // the one component lane `sub` of a chain on a whole wavefront stands for in a sweep over theta: the component whose prior term it holds, else its group's
  __device__ __forceinline__ static int sweep_comp(const unsigned char *smem, const DataRef &d, int sub) {
    return sub < d.G ? sub : (sub < d.n_obs ? (int)(smem + (size_t)64 * d.pad * 8)[sub] : -1);
  }
  static constexpr bool kDynamicLds = true;
  __host__ __device__ static size_t lds_bytes_of(const DataRef &d, int lanes, int threads) { return d.pad > 0 ? rows_lds_bytes(d.pad, threads / 64, d.G) : lds_bytes(d.n_obs, d.G, lanes); }
  __device__ static void stage(unsigned char *smem, const DataRef &d, int tid, int nt, int) {
    double *dst = reinterpret_cast<double *>(smem);
    if (d.pad > 0) {
      const int Rp = d.pad;
      for (int i = tid; i < d.n_obs; i += nt) dst[(i & 63) * Rp + (i >> 6)] = d.x[i];
      uint8_t *gd = smem + (size_t)64 * Rp * 8;
      for (int i = tid; i < 64; i += nt) gd[i] = i < d.n_obs ? d.xb[i] : 0;
      return;
    }
    uint8_t *gd = smem + (size_t)d.n_obs * 8;
    for (int i = tid; i < d.n_obs; i += nt) { dst[i] = d.x[i]; gd[i] = d.xb[i]; }
  }
  // Lane order of the priors (the same a translated closure gets, translate.js): lane 0 adds the terms outside
  // loops -- mu, sigma -- and the `for k` loop over the group means is dealt to the lanes like the data loop.
  static constexpr bool kSplitPrior = true;
  // Register mirror of the chain's state (kept current by on_set), usable when the chain's L = min(lanes, 64) lanes inside a wave are at
  // least as many as the groups and every lane meets one group only (ModelConsts::group_lane_const): lane j of the chain then holds
  // theta[j] (its prior term, its share of the range check) and theta[g[j]] (the mean of ITS observations), every lane mu and sigma --
  // an evaluation reads nothing of the state from LDS.  Round 2 read mu, sigma, theta[k], the label and theta[label] back from LDS in
  // every evaluation: four dependent round trips behind the data passes of the CU's other waves.
  static constexpr bool kTracksState = true;
  struct Cache {
    NormCache n; double th_own, th_pass, mu, sigma; int my_group; bool regs, loaded;
    // the prior of (mu, sigma) kept while neither changes (32 of the 34 updates of a step), and the constants of theta's prior held in
    // VECTOR registers: the stepper keeps more wave-uniform values alive than there are scalar registers, and every use of a spilled one is a
    // v_readlane plus a wait state -- the priors alone were 0.5 us of the 2.4 us a stepper update takes (measured by cutting them out)
    double pr_mu, pr_sigma, pr_val, c1, den1, y1h, y1l; int den1_ok;
    // lane-local re-evaluation (row layout): the last two sums this lane formed, each with what it was formed FROM -- the value the
    // sum starts with (the lane's prior terms), the mean of its observations and the sd: together they determine the sum, bit for bit
    double a_start, a_mean, a_sd, a_T, b_start, b_mean, b_sd, b_T;
    bool a_recent;
    // certified decisions (row layout; see log_post_approx below): this lane's sum of squares S2 = sum (y_i - mean)^2 over its row, with the mean it was formed for
    double s2, s2_mean;
  };
  __device__ __forceinline__ static Cache cache_init() {
    const double nan = __builtin_nan("");
    return Cache{norm_cache_init(), 0.0, 0.0, 0.0, 0.0, -1, false, false, nan, nan, 0.0, 0.0, 0.0, 0.0, 0.0, 0, nan, nan, nan, 0.0, nan, nan, nan, 0.0, false, 0.0, nan};
  }
  template <int GL>
  __device__ __forceinline__ static void load(Cache &k, const StateView &S, const ModelConsts &mc, const DataRef &d, const unsigned char *smem, int sub) {
    if (k.loaded) return;
    constexpr int L = GL < 64 ? GL : 64;
    k.loaded = true;
    k.regs = mc.group_lane_const != 0 && d.G <= L;
    const int j = sub & (L - 1);
    k.mu = S(d.G);
    k.sigma = S(d.G + 1);
    k.my_group = (k.regs && sub < d.n_obs) ? (int)(smem + (d.pad > 0 ? (size_t)64 * d.pad * 8 : (size_t)d.n_obs * 8))[sub] : -1;
    k.th_own = (k.regs && j < d.G) ? S(j) : 0.0;
    k.th_pass = k.my_group >= 0 ? S(k.my_group) : 0.0;
    k.c1 = mc.c1; k.den1 = mc.den1; k.y1h = mc.y1_hi; k.y1l = mc.y1_lo; k.den1_ok = mc.den1_ok;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(k.c1), "+v"(k.den1), "+v"(k.y1h), "+v"(k.y1l), "+v"(k.den1_ok));      // (vector registers from here on)
#endif
  }
  // prior(mu, sigma): out of line, its hyper-parameters read through the kernel-argument pointer where they are used -- two updates in 34
  __device__ inline __attribute__((noinline)) static double prior_mu_sigma_impl(double mu, double sigma, double m0, double c0, double den0, double y0h, double y0l, int den0_ok,
                                                                                 double ua, double ub, double lunif) {
    double lp = 0;
    lp += norm_const_sd(mu, m0, c0, den0, y0h, y0l, den0_ok);
    lp += (sigma < ua || sigma > ub) ? -kInf : lunif;
    return lp;
  }
  // (the kernel-argument pointer is only valid in the kernel's own body: the fields are fetched here, inlined at the rarely taken call site)
  __device__ __forceinline__ static double prior_mu_sigma_cold(double mu, double sigma) {
    const cold_args_ptr ca = cold_args();
    return prior_mu_sigma_impl(mu, sigma, ca->mc.m0, ca->mc.c0, ca->mc.den0, ca->mc.y0_hi, ca->mc.y0_lo, ca->mc.den0_ok, ca->mc.ua, ca->mc.ub, ca->mc.lunif);
  }
  __device__ __forceinline__ static void on_set(Cache &k, int comp, double v, int sub, const DataRef &d) {
    // selects, not stores under branches: the compiler merges `if (c) k.mu = v; else k.sigma = v;` into ONE store through a selected
    // address, which pins the whole cache to scratch memory (a VMEM round trip in every evaluation: round 2's 132 B of scratch)
    k.th_own = (comp == (sub & 63)) ? v : k.th_own;     // (only read when groups <= L <= 64: the lane's own index inside the chain is sub mod L)
    k.th_pass = (comp == k.my_group) ? v : k.th_pass;
    k.mu = (comp == d.G) ? v : k.mu;
    k.sigma = (comp == d.G + 1) ? v : k.sigma;
  }
  static constexpr bool kMirrorCheck = true;
  template <int GL>
  __device__ __forceinline__ static bool mirror_ok(const Cache &k, const StateView &S, const DataRef &d, int sub) {
    if (!k.loaded) return true;
    constexpr int L = GL < 64 ? GL : 64;
    const int j = sub & (L - 1);
    bool ok = f64_bits(k.mu) == f64_bits(S(d.G)) && f64_bits(k.sigma) == f64_bits(S(d.G + 1));
    if (k.regs) {
      if (j < d.G) ok = ok && f64_bits(k.th_own) == f64_bits(S(j));
      if (k.my_group >= 0) ok = ok && f64_bits(k.th_pass) == f64_bits(S(k.my_group));
    }
    return ok;
  }
  template <class C>
  __device__ __forceinline__ static double prior(const StateView &, const ModelConsts &mc, const DataRef &, C &k) {
    const double mu = k.mu, sigma = k.sigma;
#if defined(__HIP_DEVICE_COMPILE__)
    if (mu != k.pr_mu || sigma != k.pr_sigma) { k.pr_mu = mu; k.pr_sigma = sigma; k.pr_val = prior_mu_sigma_cold(mu, sigma); }      // (NaN keys never match)
    return k.pr_val;
#else
    double lp = 0;
    lp += norm_const_sd(mu, mc.m0, mc.c0, mc.den0, mc.y0_hi, mc.y0_lo, mc.den0_ok);
    lp += (sigma < mc.ua || sigma > mc.ub) ? -kInf : mc.lunif;
    return lp;
#endif
  }
  template <int G, class C>
  __device__ __forceinline__ static double prior_split(const StateView &S, const ModelConsts &mc, const DataRef &d, int sub, double acc, C &k) {
    const double mu = k.mu;
    if (k.regs) {      // groups <= lanes: at most one term per lane, theta[sub] from the mirror
      if (sub < d.G) acc += norm_const_sd(k.th_own, mu, k.c1, k.den1, k.y1h, k.y1l, k.den1_ok);
      return acc;
    }
    for (int q = sub; q < d.G; q += G) acc += norm_const_sd(S(q), mu, mc.c1, mc.den1, mc.y1_hi, mc.y1_lo, mc.den1_ok);
    return acc;
  }
  template <int GL>
  __device__ __forceinline__ static Pass begin(const StateView &S, const ModelConsts &mc, const DataRef &d,
                                               const unsigned char *smem, Cache &k) {
    Pass ps;
    NormCache &kc = k.n;
    norm_cache_update<true>(kc, k.sigma, mc.neg_half_log_2pi);
    ps.c = kc.c;
    ps.den = kc.den;
    ps.y = kc.y;
    bool ok = !mc.exact_division && mc.data_mid_range && kc.den_ok;
    // every group mean must be inside the range amwg_div.h needs: the L lanes that run this chain inside the wave check L means
    // at a time and agree through a ballot (round 1 had every lane walk all the means: 32 dependent LDS reads per evaluation)
    {
      constexpr int L = GL < 64 ? GL : 64;
      const int lane = (int)(threadIdx.x & 63u), sub_in_wave = lane & (L - 1);
      bool mine = true;
#if defined(AMWG_X_CUT_BEGIN)
      if (k.regs) { }
#else
      if (k.regs) { if (sub_in_wave < d.G) { const double th = k.th_own; mine = th == 0 || mid_range(__builtin_fabs(th)); } }
#endif
      else for (int q = sub_in_wave; q < d.G; q += L) { const double th = S(q); mine = mine && (th == 0 || mid_range(__builtin_fabs(th))); }
      if constexpr (L == 1) ok = ok && mine;
      else {
        const uint64_t all = __ballot(mine);
        const uint64_t group = (L == 64 ? ~0ull : ((1ull << (L & 63)) - 1ull)) << (lane & ~(L - 1));
        ok = ok && ((all & group) == group);
      }
    }
    ps.fast = ok;
    ps.x = reinterpret_cast<const double *>(smem);
    ps.g = smem + (size_t)d.n_obs * 8;
    ps.S = S;
    ps.lane_const = mc.group_lane_const != 0;
    ps.regs = k.regs;
    ps.rows = d.pad > 0;
    ps.th_pass = k.th_pass;
    return ps;
  }
  // ---------------------------------------------------------------------------------------------------------------------------------
  // LANE-LOCAL RE-EVALUATION (row layout, a chain on one whole wavefront).  mcmc.js:524-526 evaluates the whole log_post for every update.
  // In the 64-lane order log_post is the butterfly of 64 per-lane sums, and the sum of lane j is a pure function of three numbers: the value
  // it starts from (the lane's prior terms), the mean of its observations (its ONE group's theta) and sd.  An update of theta_g changes those
  // for the lanes of group g only (two of 64 in cfg4), so every other lane's sum is -- bit for bit -- the one it formed last time.  Each lane
  // keeps its last two sums with what they were formed from (two: the committed state and the proposal under evaluation; the less recently
  // used one is replaced), and only the lanes whose three numbers match neither are re-formed: all 64 lanes compute the terms of such a
  // lane's observations side by side (read from its row of the tile), leave them in LDS, and the lane itself adds them up IN ORDER --
  //     sum = start; for r = 0, 1, ...: sum += term_r            the same additions of the same values in the same order as the ordinary pass
  // -- ~160 dependent additions on one lane (latency the SIMD's other wave fills) instead of ~1 270 instructions on all 64.  mu and sigma change
  // every lane's numbers: then, and whenever more than kMaxLocal lanes are stale, the ordinary pass runs.  Results are IDENTICAL to evaluating
  // everything (tests: every 64-lane comparison with the oracle runs through this; options.full_evaluation = 1 switches it off, and the two are
  // compared chain by chain), so -- like the cached log_post of the current state -- it is not an approximation but work not done twice.
  static constexpr bool kLaneReuse = true;
  template <int U>
  __device__ __forceinline__ static double rows_full(const Pass &ps, const double *row, int n_obs, int sub, double acc) {
    const int n_full = n_obs >> 6, rem = n_obs & 63;
    const double last = row[sub < rem ? n_full : 0];      // (the remainder round's observation, requested before the pass)
    if (ps.fast) {
      acc = norm_pass_staged<1, U, false>(row, nullptr, StateView{nullptr}, ps.th_pass, ps.c, ps.den, ps.y, n_full, 0, acc);
      const double t = last - ps.th_pass;
      const double term = ps.c - div_by_invariant(t * t, ps.den, ps.y);
      return sub < rem ? acc + term : acc;
    }
    for (int r = 0; r < n_full; ++r) { const double t = row[r] - ps.th_pass; acc += ps.c - (t * t) / ps.den; }
    const double t = last - ps.th_pass;
    const double term = ps.c - (t * t) / ps.den;
    return sub < rem ? acc + term : acc;
  }
  template <int U>
  __device__ __forceinline__ static double lane_sum_rows(Cache &k, const Pass &ps, double start, int n_obs, int groups, int sub, const unsigned char *smem, int pitch, int wave) {
    const double *tile = reinterpret_cast<const double *>(smem);
    const int rows = local_rows(groups);
    const int spitch = term_pitch(pitch);
    double *scratch = const_cast<double *>(tile) + (size_t)64 * pitch + 8 + (size_t)wave * rows * spitch;      // (+ 8 doubles: the 64 label bytes)
    const double mean = ps.th_pass, sd = k.n.sd;
    const bool hitA = f64_bits(start) == f64_bits(k.a_start) && f64_bits(mean) == f64_bits(k.a_mean) && f64_bits(sd) == f64_bits(k.a_sd);
    const bool hitB = f64_bits(start) == f64_bits(k.b_start) && f64_bits(mean) == f64_bits(k.b_mean) && f64_bits(sd) == f64_bits(k.b_sd);
    const bool miss = !(hitA || hitB);
    const uint64_t missing = __ballot(miss);
    double T = hitA ? k.a_T : k.b_T;
    if (missing != 0ull) {
      double Tn;
      if (!ps.fast || __popcll(missing) > rows) {
        Tn = rows_full<U>(ps, tile + (size_t)sub * pitch, n_obs, sub, start);
      } else {
        const int n_full = n_obs >> 6, rem = n_obs & 63;
        // -- the terms of the stale lanes' observations, side by side: two lanes (= the two lanes of a group in cfg4) per trip, two rounds of
        // 64 observations each in flight -- four LDS reads requested before the first subtraction
        uint64_t m = missing;
        int q = 0, my_slot = 0;
        while (m != 0ull) {      // (scalar loop over the stale lanes: at most kMaxLocal)
          const int o0 = __builtin_ctzll(m);
          m &= m - 1ull;
          const bool two = m != 0ull;
          const int o1 = two ? __builtin_ctzll(m) : o0;
          if (two) m &= m - 1ull;
          const double mean0 = lane_double(mean, o0), mean1 = lane_double(mean, o1);
          const int n0 = n_full + (o0 < rem ? 1 : 0), n1 = two ? n_full + (o1 < rem ? 1 : 0) : 0;
          const int n_hi = n0 > n1 ? n0 : n1;
          const double *row0 = tile + (size_t)o0 * pitch, *row1 = tile + (size_t)o1 * pitch;
          double *out0 = scratch + (size_t)q * spitch, *out1 = scratch + (size_t)(q + 1) * spitch;
          for (int r = sub; r < n_hi; r += 128) {
            const int ra = r, rb = r + 64;
            const int ca = ra < pitch ? ra : 0, cb = rb < pitch ? rb : 0;      // (reads past a row's end: any valid address, not stored)
            const double x0a = row0[ca], x1a = row1[ca], x0b = row0[cb], x1b = row1[cb];
            AMWG_STAGE_FENCE();
            const double t0a = x0a - mean0, t1a = x1a - mean1, t0b = x0b - mean0, t1b = x1b - mean1;
            const double e0a = ps.c - div_by_invariant(t0a * t0a, ps.den, ps.y), e1a = ps.c - div_by_invariant(t1a * t1a, ps.den, ps.y);
            const double e0b = ps.c - div_by_invariant(t0b * t0b, ps.den, ps.y), e1b = ps.c - div_by_invariant(t1b * t1b, ps.den, ps.y);
            if (ra < n0) out0[ra] = e0a;
            if (ra < n1) out1[ra] = e1a;
            if (rb < n0) out0[rb] = e0b;
            if (rb < n1) out1[rb] = e1b;
          }
          my_slot = sub == o0 ? q : (two && sub == o1 ? q + 1 : my_slot);
          q += 2;
        }
        AMWG_STAGE_FENCE();
        Tn = start;
        if (miss) {      // the stale lanes add their terms up, in order: sixteen per trip, the second eight requested before the first eight are added
          const double *mine = scratch + (size_t)my_slot * spitch;
          const int n_me = n_full + (sub < rem ? 1 : 0);
          int r = 0;
          if (n_me >= 16) {
            double va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) va[u] = mine[u];
            for (; r + 16 <= n_me; r += 16) {
#pragma unroll
              for (int u = 0; u < 8; ++u) vb[u] = mine[r + 8 + u];
              AMWG_STAGE_FENCE();
#pragma unroll
              for (int u = 0; u < 8; ++u) Tn = Tn + va[u];
#pragma unroll
              for (int u = 0; u < 8; ++u) va[u] = mine[r + 16 + u];      // (a row has 16 spare slots behind its last term: read, never added)
              AMWG_STAGE_FENCE();
#pragma unroll
              for (int u = 0; u < 8; ++u) Tn = Tn + vb[u];
            }
          }
          for (; r < n_me; ++r) Tn = Tn + mine[r];
        }
      }
      if (miss) {
        T = Tn;
        if (k.a_recent) { k.b_start = start; k.b_mean = mean; k.b_sd = sd; k.b_T = Tn; k.a_recent = false; }
        else { k.a_start = start; k.a_mean = mean; k.a_sd = sd; k.a_T = Tn; k.a_recent = true; }
      }
    }
    if (!miss) k.a_recent = hitA;
    return T;
  }
  // SWEEP PREFETCH.  The updates of theta's components within one step draw their proposals from the chain's stream one after the other, and
  // nothing they draw depends on what was accepted before (a proposal is rnorm(theta_c, sd_c): the component's OWN value and scale, which only its
  // own update changes; the accept uniform follows it in the stream whatever log_post returns).  The stepper therefore draws the proposals of the
  // whole sweep when it begins (same uniforms for the same purposes in the same order), and hands them over here -- lane c holds the proposal of
  // theta_c --: every lane forms, in ONE ordinary pass over its own row, the sum it will be asked for when ITS group's update comes (its start
  // value with the proposed theta in its prior term, the proposed mean, the current sd), and leaves it in the cache entry that does not hold the
  // committed state's sum.  The 32 updates then find every lane's sum in the cache: a step makes three passes (mu, sigma, this one) instead of
  // two passes and 32 re-formations of ~160 dependent additions each.  Nothing is approximated: an entry is only ever used when its three
  // numbers match bit for bit, and whatever is not found is formed as before.
  // -> what the stepper's sweep needs: ok (wave-uniform: everything below holds and the sums are there), this lane's committed sum T_cur and its sum under
  // the proposal T_new, comp = the ONE component this lane's sum depends on (its group's mean; for the lanes that hold a term of theta's prior the
  // same component: checked) or -1.  !ok: nothing was stored that a later evaluation could not use; the stepper goes on update by update.
  struct SweepRows { bool ok; double T_cur, T_new; int comp; bool new_in_b; double mean_new; };
  template <int U>
  __device__ __forceinline__ static SweepRows prefetch_rows(Cache &k, const StateView &S, const ModelConsts &mc, const DataRef &d, const unsigned char *smem, int sub, double prop_own, int pitch) {
    SweepRows out{false, 0.0, 0.0, -1, false, 0.0};
#if defined(__HIP_DEVICE_COMPILE__)
    load<64>(k, S, mc, d, smem, sub);
    if (!k.regs) return out;      // (wave-uniform)
    Pass ps = begin<64>(S, mc, d, smem, k);
    const double sd = k.n.sd;
    const double own = sub < d.G ? prop_own : k.th_own;
    double mean = 0.0;
    {
      const int src = (k.my_group >= 0 ? k.my_group : sub) << 2;
      const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(f64_bits(prop_own) >> 32)), lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)f64_bits(prop_own));
      mean = k.my_group >= 0 ? bits_f64(((uint64_t)hi << 32) | (uint64_t)lo) : 0.0;
    }
    // the passes below are the fast one: every proposed mean inside the range amwg_div.h needs (a lane checks the component it holds), like begin() checks
    // the current ones; and a lane that holds a term of theta's prior must have that same component as its group (its sum then depends on ONE component)
    const bool mine_ok = !(sub < d.G) || own == 0 || mid_range(__builtin_fabs(own));
    const bool one_comp = !(sub < d.G && k.my_group >= 0 && k.my_group != sub);
    if (!ps.fast || __ballot(mine_ok && one_comp) != ~0ull) return out;
    // this lane's start value as log_post forms it (kernel: prior on lane 0, then the lane's term of theta's prior), now and under the proposal
    const double pr = prior(S, mc, d, k);
    Cache k2 = k;
    k2.th_own = own;
    const double start_new = prior_split<64>(S, mc, d, sub, (sub == 0) ? pr : 0.0, k2);
    const double start_cur = prior_split<64>(S, mc, d, sub, (sub == 0) ? pr : 0.0, k);
    const double *row = reinterpret_cast<const double *>(smem) + (size_t)sub * pitch;
    // the committed state's sum: in the cache (the usual case), else formed now -- by all lanes, into entry a (the first sweep of a launch)
    bool curA = f64_bits(start_cur) == f64_bits(k.a_start) && f64_bits(k.th_pass) == f64_bits(k.a_mean) && f64_bits(sd) == f64_bits(k.a_sd);
    const bool curB = f64_bits(start_cur) == f64_bits(k.b_start) && f64_bits(k.th_pass) == f64_bits(k.b_mean) && f64_bits(sd) == f64_bits(k.b_sd);
    if (__ballot(!(curA || curB)) != 0ull) {
      const double Tc = rows_full<U>(ps, row, d.n_obs, sub, start_cur);
      if (!(curA || curB)) { k.a_start = start_cur; k.a_mean = k.th_pass; k.a_sd = sd; k.a_T = Tc; curA = true; }
    }
    out.T_cur = curA ? k.a_T : k.b_T;
    ps.th_pass = mean;
    const double Tn = rows_full<U>(ps, row, d.n_obs, sub, start_new);
    const bool intoB = curA;
    k.b_start = intoB ? start_new : k.b_start; k.b_mean = intoB ? mean : k.b_mean; k.b_sd = intoB ? sd : k.b_sd; k.b_T = intoB ? Tn : k.b_T;
    k.a_start = intoB ? k.a_start : start_new; k.a_mean = intoB ? k.a_mean : mean; k.a_sd = intoB ? k.a_sd : sd; k.a_T = intoB ? k.a_T : Tn;
    k.a_recent = intoB;      // (the committed one counts as recently used: a miss replaces the other)
    out.ok = true;
    out.T_new = Tn;
    out.comp = k.my_group >= 0 ? k.my_group : (sub < d.G ? sub : -1);
    out.new_in_b = intoB;
    out.mean_new = mean;
#endif
    return out;
  }
  // CERTIFIED DECISIONS in the row layout (the sweep kernel; amwg_kernel.h).  A lane's sum as a real number is
  //     start + n_l c - S2 / den,      S2 = sum over its row of (y_i - mean)^2
  // and S2 depends on the lane's MEAN only: neither sigma's nor mu's update needs a pass over the data (c, den and the start values are a handful of
  // operations), and a sweep over theta needs the S2 of the proposed means -- two operations per observation (sub, fma) instead of the eight of the term.
  // Bounds, u = 2^-53, per lane m_l = |start| + n_l |c| + Q_l (Q_l = S2 / den >= 0), M = their sum over the wavefront: the expression's lane sum (start, then n_l
  // terms c - RN(tt / den) added in order) is within (n_l + 2) u m_l of the real number, the value here within (n_l / 4 + 9) u m_l (four partial sums of non-negative
  // terms, exact squares inside the fma, 1 / den correctly rounded, n_l c and two additions), each butterfly within 6 u M:
  // -- that much is the distance to the expression summed in the chain's 64-lane order.  What the bounds below are FOR is the reference's own order (round 5,
  // last part): one running sum, prior terms first, then the observations i = 0, 1, ... (reference_order below, the expression these kernels evaluate when a
  // uniform falls inside a bound, and the value a launch leaves behind).  Its n + G + 2 additions each round a partial sum that stays below
  // M' = M + 2 |prior(mu, sigma)| + 2 |lunif| (lane 0's start value is the rounded sum of three prior terms: their magnitudes add up to no more than that), its n
  // terms carry 5 u (n |c| + Q) between them: within (n + G + 8) u M' of the real number.  With the (n_l / 4 + 15) u M of the value here:
  //     a value of log_post:      eps = u M' (2 (n + G) + 64) 1.25
  //     a difference of two (the sweep's D_c, which also leaves out the lanes that do not change):   eps = u M' (4 (n + G) + 128) 1.25,  M' over max(m_l, m_l')
  // -- a decision certified with these is the one the REFERENCE makes (one lane per chain), not merely the one this geometry's own summation order would
  // make: accept counts are the reference's at 64 lanes per chain (tests/test_gpu_decision_parity.py: zero first flips where rounds 3-4 counted ~0.1 per 1e9).
  static constexpr bool kCertified = true;
  static constexpr int kCertifiedLanes = 64;
  static constexpr bool kCertifiedNeedsRows = true;      // (only the sweep kernel: the row tile is what the S2 pass reads)
  struct Approx { double value, eps; };
  // S2 of this lane's row for `mean`: four interleaved partial sums (the order is free: the value is used with its bound)
  __device__ __forceinline__ static double rows_sq(const double *row, double mean, int n_obs, int sub) {
    const int n_full = n_obs >> 6, rem = n_obs & 63;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    auto eight = [&](const double (&x)[8]) {
      { const double t = x[0] - mean; a0 = __builtin_fma(t, t, a0); }
      { const double t = x[1] - mean; a1 = __builtin_fma(t, t, a1); }
      { const double t = x[2] - mean; a2 = __builtin_fma(t, t, a2); }
      { const double t = x[3] - mean; a3 = __builtin_fma(t, t, a3); }
      { const double t = x[4] - mean; a0 = __builtin_fma(t, t, a0); }
      { const double t = x[5] - mean; a1 = __builtin_fma(t, t, a1); }
      { const double t = x[6] - mean; a2 = __builtin_fma(t, t, a2); }
      { const double t = x[7] - mean; a3 = __builtin_fma(t, t, a3); }
    };
    // (a version with two register sets -- the next eight observations requested before this trip's arithmetic -- was measured in round 6: the pass's share of a step
    // fell from 6 500 to 5 400 cycles under tools/phase_clock.py, the launch time did not move (1.65 against 1.66 ms per 100 steps: the SIMD's other wavefront fills
    // the wait), and the sixteen extra registers cost this kernel 17 more spills.  One set.)
    int r = 0;
    for (; r + 8 <= n_full; r += 8) {
      double x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = row[r + u];
      AMWG_STAGE_FENCE();
      eight(x);
    }
    for (; r < n_full; ++r) { const double t = row[r] - mean; a0 = __builtin_fma(t, t, a0); }
    if (sub < rem) { const double t = row[n_full] - mean; a1 = __builtin_fma(t, t, a1); }
    return (a0 + a1) + (a2 + a3);
  }
  // ... the committed mean's S2, from the cache when the mean has not changed (every lane takes part: a wavefront-uniform call)
  __device__ __forceinline__ static double lane_s2(Cache &k, const unsigned char *smem, double mean, const DataRef &d, int sub) {
    const bool stale = f64_bits(mean) != f64_bits(k.s2_mean);
    if (__ballot(stale) != 0ull) {
      const double v = rows_sq(reinterpret_cast<const double *>(smem) + (size_t)sub * d.pad, mean, d.n_obs, sub);
      if (stale) { k.s2 = v; k.s2_mean = mean; }
    }
    return k.s2;
  }
  struct ApproxLane { double value, mag; };      // this lane's start + n_l c - Q and |start| + n_l |c| + Q
  __device__ __forceinline__ static ApproxLane approx_lane(const Cache &k, double start, double s2, const DataRef &d, int sub) {
    const double n_l = (double)((d.n_obs >> 6) + (sub < (d.n_obs & 63) ? 1 : 0));
    const double q = s2 * k.n.y.hi, nc = n_l * k.n.c;
    return ApproxLane{(start + nc) - q, __builtin_fabs(start) + __builtin_fabs(nc) + q};
  }
  __device__ __forceinline__ static double value_bound(double M, const DataRef &d) { return M * (2.0 * (double)(d.n_obs + d.G) + 64.0) * 1.25 * 0x1p-53; }
  __device__ __forceinline__ static double difference_bound(double M, const DataRef &d) { return M * (4.0 * (double)(d.n_obs + d.G) + 128.0) * 1.25 * 0x1p-53; }
  // (lane 0's share of M': see above)
  __device__ __forceinline__ static double prior_magnitude(double pr, const ModelConsts &mc, int sub) { return sub == 0 ? 2.0 * (__builtin_fabs(pr) + __builtin_fabs(mc.lunif)) : 0.0; }
  // THE REFERENCE'S ORDER in the row layout: log_post as the reference's closure forms it -- lp = prior(mu) + prior(sigma); for g: lp += prior(theta_g); for i: lp +=
  // term_i (mcmc.js:524-526 calls it, the model is tests/model_spec.py's closure) -- with every lane computing the terms of ITS observations (the same operations
  // on the same values as the one-lane kernel: a term depends on its observation, its group's mean and sigma only) and ONE running sum that visits them in the
  // order i = 64 r + lane.  ~1e4 dependent additions through v_readlane: 200 us, which is why it is not the pass -- it runs when a uniform falls inside a
  // certified bound (some 5e-8 of the updates at cfg4) and when the host asks for log_post.  Wave-uniform result; every lane of the wavefront takes part.
  static constexpr bool kReferenceOrder = true;
  __device__ inline __attribute__((noinline)) static double reference_order_sum(double acc, const double *state, int G, double mu, double c1, double den1, double y1h, double y1l, int den1_ok,
                                                                                const double *row, int n_obs, int sub, double mean, double c, double den, double yh, double yl, bool fast) {
#if defined(__HIP_DEVICE_COMPILE__)
    for (int q = 0; q < G; ++q) acc += norm_const_sd(state[q], mu, c1, den1, y1h, y1l, den1_ok);
    const int n_full = n_obs >> 6, rem = n_obs & 63;
    for (int r = 0; r <= n_full; ++r) {
      const int cnt = r < n_full ? 64 : rem;
      const double t = row[sub < cnt ? r : 0] - mean;      // (a lane without an observation in the last round: any valid address, its term is not added)
      const double tt = t * t;
      const double term = c - (fast ? div_by_invariant(tt, den, Reciprocal{yh, yl}) : tt / den);
      for (int l = 0; l < cnt; ++l) acc += lane_double(term, l);
    }
#endif
    return acc;
  }
  template <int G>
  __device__ __forceinline__ static double reference_order(Cache &k, const StateView &S, const ModelConsts &mc, const DataRef &d, const unsigned char *smem, int sub) {
    static_assert(G == 64, "the row layout: a chain on one wavefront");
    load<G>(k, S, mc, d, smem, sub);
    const Pass ps = begin<G>(S, mc, d, smem, k);
    const double pr = prior(S, mc, d, k);
    const double mean = sub < d.n_obs ? S((int)(smem + (size_t)64 * d.pad * 8)[sub]) : 0.0;      // this lane's ONE group (group_lane_const: what the row layout is chosen for)
    return reference_order_sum(pr, S.base, d.G, k.mu, k.c1, k.den1, k.y1h, k.y1l, k.den1_ok, reinterpret_cast<const double *>(smem) + (size_t)sub * d.pad, d.n_obs, sub, mean,
                               ps.c, ps.den, ps.y.hi, ps.y.lo, ps.fast);
  }
  // log_post of the state as it stands (the stepper has stored its proposal), cheaply: mu's and sigma's updates (no pass: every lane's mean is the cached one),
  // any other update of the ordinary stepper (the lanes whose mean changed re-form their S2)
  template <int G, int BT>
  __device__ __forceinline__ static Approx log_post_approx(Cache &k, const StateView &S, const ModelConsts &mc, const DataRef &d, const unsigned char *smem, int sub) {
#if defined(__HIP_DEVICE_COMPILE__)
    load<G>(k, S, mc, d, smem, sub);
    if (!k.regs || d.pad <= 0) return Approx{0.0, __builtin_inf()};      // (wave-uniform: not the row layout with the register mirror -- the expression)
    norm_cache_update<true>(k.n, k.sigma, mc.neg_half_log_2pi);
    const double pr = prior(S, mc, d, k);
    const double start = start_value(k, k.mu, pr, sub, d);
    const double s2 = lane_s2(k, smem, k.th_pass, d, sub);
    const ApproxLane a = approx_lane(k, start, s2, d, sub);
    const double value = butterfly<1, 64>(a.value), M = butterfly<1, 64>(a.mag + prior_magnitude(pr, mc, sub));
    return Approx{value, value_bound(M, d)};
#else
    return Approx{0.0, __builtin_inf()};
#endif
  }
  // the sweep: every lane's value now and under its entry's proposal (lane c < groups holds the proposal of theta_c, as for prefetch_rows)
  struct SweepApprox { bool ok; int comp; double cur, neu, mag, mean_new, s2_new; };
  __device__ __forceinline__ static SweepApprox sweep_approx(Cache &k, const StateView &S, const ModelConsts &mc, const DataRef &d, const unsigned char *smem, int sub, double prop_own) {
    SweepApprox out{false, -1, 0.0, 0.0, 0.0, 0.0, 0.0};
#if defined(__HIP_DEVICE_COMPILE__)
    load<64>(k, S, mc, d, smem, sub);
    if (!k.regs || d.pad <= 0) return out;      // (wave-uniform)
    norm_cache_update<true>(k.n, k.sigma, mc.neg_half_log_2pi);
    const double own = sub < d.G ? prop_own : k.th_own;
    double mean = 0.0;
    {
      const int src = (k.my_group >= 0 ? k.my_group : sub) << 2;
      const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(f64_bits(prop_own) >> 32)), lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)f64_bits(prop_own));
      mean = k.my_group >= 0 ? bits_f64(((uint64_t)hi << 32) | (uint64_t)lo) : 0.0;
    }
    // (a lane that holds a term of theta's prior must have that same entry as its group: its value then depends on ONE entry)
    const bool one_comp = !(sub < d.G && k.my_group >= 0 && k.my_group != sub);
    if (__ballot(one_comp) != ~0ull) return out;
    const double pr = prior(S, mc, d, k);
    Cache k2 = k;
    k2.th_own = own;
    const double start_new = prior_split<64>(S, mc, d, sub, (sub == 0) ? pr : 0.0, k2);
    const double start_cur = prior_split<64>(S, mc, d, sub, (sub == 0) ? pr : 0.0, k);
    const double s2_cur = lane_s2(k, smem, k.th_pass, d, sub);
    const double s2_new = rows_sq(reinterpret_cast<const double *>(smem) + (size_t)sub * d.pad, mean, d.n_obs, sub);
    const ApproxLane c0 = approx_lane(k, start_cur, s2_cur, d, sub), c1 = approx_lane(k, start_new, s2_new, d, sub);
    out.ok = true;
    out.comp = k.my_group >= 0 ? k.my_group : (sub < d.G ? sub : -1);
    out.cur = c0.value; out.neu = c1.value;
    out.mag = (c0.mag > c1.mag ? c0.mag : c1.mag) + prior_magnitude(pr, mc, sub);
    out.mean_new = mean; out.s2_new = s2_new;
#endif
    return out;
  }
  // ... and what the accepted entries leave behind in this lane: the register mirror (as sweep_commit_all) and the S2 that goes with the new mean
  __device__ __forceinline__ static void sweep_approx_commit(Cache &k, const SweepApprox &sa, uint64_t acc_mask, bool mine, double prop_own, int sub, const DataRef &d) {
    const bool own = sub < d.G && ((acc_mask >> (sub & 63)) & 1ull) != 0ull;
    k.th_own = own ? prop_own : k.th_own;
    const bool grp = k.my_group >= 0 && mine;
    k.th_pass = grp ? sa.mean_new : k.th_pass;
    k.s2 = grp ? sa.s2_new : k.s2;
    k.s2_mean = grp ? sa.mean_new : k.s2_mean;
  }
  __device__ __forceinline__ static double start_value(const Cache &k, double mu, double pr, int sub, const DataRef &d) {      // prior_split for the register mirror, mu given
    double acc = (sub == 0) ? pr : 0.0;
    if (sub < d.G) acc += norm_const_sd(k.th_own, mu, k.c1, k.den1, k.y1h, k.y1l, k.den1_ok);
    return acc;
  }
  // the accepted entries of a sweep decided all at once (amwg_kernel.h): what on_set does for one store, for every accepted entry -- bit c of acc_mask: entry c
  __device__ __forceinline__ static void sweep_commit_all(Cache &k, const SweepRows &r, uint64_t acc_mask, double prop_own, int sub, const DataRef &d) {
    const bool own = sub < d.G && ((acc_mask >> (sub & 63)) & 1ull) != 0ull;
    k.th_own = own ? prop_own : k.th_own;
    const bool grp = k.my_group >= 0 && ((acc_mask >> (k.my_group & 63)) & 1ull) != 0ull;
    k.th_pass = grp ? r.mean_new : k.th_pass;
  }
  // after the sweep: the entry that now holds the committed state's sum is the recently used one (the lane's component accepted: the proposal's)
  __device__ __forceinline__ static void sweep_done(Cache &k, const SweepRows &r, bool accepted_mine) { k.a_recent = r.new_in_b != accepted_mine; }
  __device__ __forceinline__ static double lane_double(double v, int src) {      // v of lane `src` (wave-uniform)
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(f64_bits(v) >> 32), src);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)f64_bits(v), src);
    return bits_f64(((uint64_t)hi << 32) | (uint64_t)lo);
#else
    return v;
#endif
  }

  template <bool FAST>
  __device__ __forceinline__ static double term(const Pass &ps, int i) {
    const double t = ps.x[i] - ps.S(ps.g[i]);
    const double tt = t * t;
    return ps.c - (FAST ? div_by_invariant(tt, ps.den, ps.y) : tt / ps.den);
  }
  template <int G>
  __device__ inline __attribute__((noinline)) static double pass_slow_impl(const double *x, const uint8_t *g, const StateView S, double c, double den, int n_obs, int sub, double acc) {
    for (int i = sub; i < n_obs; i += G) { const double t = x[i] - S(g[i]); acc += c - (t * t) / den; }
    return acc;
  }
  template <int G>
  __device__ __forceinline__ static double pass_slow(const Pass &ps, int n_obs, int sub, double acc) { return pass_slow_impl<G>(ps.x, ps.g, ps.S, ps.c, ps.den, n_obs, sub, acc); }
  // the fast pass, hand-pipelined: group indices two blocks ahead, y and theta[g] one block ahead of the arithmetic
  static constexpr bool kStagedFast = true;
  // When the group labels repeat with the lane stride (g[i] == g[i % G], checked once on the host: ModelConsts::group_lane_const --
  // e.g. the balanced design g_i = i mod 32 on 64 lanes) a lane meets one group only: its mean is read once and the pass is the
  // constant-mean one (no index reads, no gather: 8.1 instead of 9.4 VALU and 1 instead of 2.5 LDS reads per observation -- the
  // gathered pass keeps the LDS pipe of a CU ~95 % busy).  Same operations on the same values in the same order either way.
  template <int G, int U = 8>
  __device__ __forceinline__ static double pass_fast(const Pass &ps, int n_obs, int sub, double acc) {
    if (ps.lane_const) {
      const double m = ps.regs ? ps.th_pass : (sub < n_obs ? ps.S(ps.g[sub]) : 0.0);
      return norm_pass_staged<G, U, false>(ps.x, nullptr, ps.S, m, ps.c, ps.den, ps.y, n_obs, sub, acc);
    }
    return pass_gathered<G>(ps.x, ps.g, ps.S, ps.c, ps.den, ps.y, n_obs, sub, acc);
  }
  // (out of line: a design whose labels do not repeat with the lane stride runs this one instead of the constant-mean pass -- never both)
  template <int G>
  __device__ inline __attribute__((noinline)) static double pass_gathered(const double *x, const uint8_t *g, const StateView S, double c, double den, Reciprocal y,
                                                                         int n_obs, int sub, double acc) {
    return norm_pass_staged<G, 4, true>(x, g, S, 0.0, c, den, y, n_obs, sub, acc);
  }
};

// ---------------------------------------------------------------------------------------------
// y_i ~ pois(exp(sum_k X[i][k]*beta[k] + [i >= cp]*beta[7])); beta_k ~ norm(0,10); cp ~ unif(0,N-1)
// components: beta[0..7], cp (int)                                        SURVEY.md §8(d) cfg5
// The design matrix (3.6 MB at N=5e4) is read straight from L2/MALL, stored column-major so the
// G lanes of a chain read consecutive observations of one column (coalesced); the per-observation
// exp+log dominate the arithmetic by two orders of magnitude.
struct PoisGlmModel {
  static constexpr bool kUser = false, kHasFast = false, kOneLanePass = false;
  static constexpr int kDerived = 0;
  static constexpr bool kHasBinary = false;   // real / int parameters only: the BinaryStepper branch is not compiled in
  static constexpr int kMaxThreads = 256;    // exp+log per observation want > 128 VGPRs (capped at 128 the kernel spills 73 of them; round 3 measured it 1 % faster all the same: VALU-issue bound); no LDS tile to share anyway
  static constexpr int kUnroll = 2;   // exp+log per term: more would spill
  // col[k]: column k of the design matrix, then y and lfactorial(y) -- nine wave-uniform base pointers (scalar registers); an
  // observation is addressed by ONE 32-bit byte offset per lane (global_load ... vOffset, sBase) instead of nine 64-bit adds
  // icp: the change point as an integer threshold (observation i gets b[7] iff i >= icp, see begin()); K: the constants of exp / log
  // in vector registers (the scalar ones of this kernel are better spent on the nine base pointers, which must be scalar)
  struct Pass { double b[8]; double cp; int icp; ExpLogRegs K; const char *col[9]; };
  __host__ __device__ static size_t lds_bytes(int, int, int) { return 0; }
  __device__ static void stage(unsigned char *, const DataRef &, int, int, int) {}
  // closure order: `for k` over the 8 coefficients (dealt to the lanes), then the change point's prior (lane 0), then the data
  static constexpr bool kSplitPrior = true;
  template <class C>
  __device__ __forceinline__ static double prior(const StateView &, const ModelConsts &, const DataRef &, C &) { return 0.0; }
  template <int G, class C>
  __device__ __forceinline__ static double prior_split(const StateView &S, const ModelConsts &mc, const DataRef &, int sub, double acc, C &) {
    for (int k = sub; k < 8; k += G) acc += norm_const_sd(S(k), mc.m0, mc.c0, mc.den0, mc.y0_hi, mc.y0_lo, mc.den0_ok);
    if (sub == 0) {
      const double cp = S(8);
      acc += (cp < 0 || cp > mc.cp_upper) ? -kInf : mc.lunif_cp;
    }
    return acc;
  }
  struct Cache {};
  __device__ __forceinline__ static Cache cache_init() { return Cache{}; }
  template <int GL>
  __device__ __forceinline__ static Pass begin(const StateView &S, const ModelConsts &, const DataRef &d,
                                               const unsigned char *, Cache &) {
    Pass ps;
#pragma unroll
    for (int k = 0; k < 8; ++k) ps.b[k] = S(k);
    ps.cp = S(8);
    // `i >= state.cp` (i = 0, 1, ...; both JavaScript numbers) as an integer comparison: i >= cp <=> i >= ceil(cp); a NaN never
    // compares (threshold beyond every index: n_obs <= 2^28, amwg_create), anything <= 0 always does
    ps.icp = !(ps.cp < 536870912.0) ? 536870912 : (ps.cp <= 0.0 ? 0 : (int)__builtin_ceil(ps.cp));
    ps.K = exp_log_regs();
#pragma unroll
    for (int k = 0; k < 7; ++k) ps.col[k] = reinterpret_cast<const char *>(d.x + (size_t)k * (size_t)d.n_obs);
    ps.col[7] = reinterpret_cast<const char *>(d.y);
    ps.col[8] = reinterpret_cast<const char *>(d.lfact);
    return ps;
  }
  struct Row { double v[9]; };
  __device__ __forceinline__ static void load_row(const Pass &ps, int i, Row &r) {
    const uint32_t off = (uint32_t)i * 8u;     // n_obs <= 2^28 (amwg_create)
#pragma unroll
    for (int k = 0; k < 9; ++k) r.v[k] = *reinterpret_cast<const double *>(ps.col[k] + off);
  }
  // mode (wave-uniform): 0 = i < icp for every lane of the wave, 2 = i >= icp for every lane, 1 = compare
  __device__ __forceinline__ static double eta_of(const Pass &ps, const Row &r, int i, int mode) {
    double eta = r.v[0] * ps.b[0];
#pragma unroll
    for (int k = 1; k < 7; ++k) eta += r.v[k] * ps.b[k];
    if (mode == 2) eta += ps.b[7];
    else if (mode == 1) { if (i >= ps.icp) eta += ps.b[7]; }
    return eta;
  }
  __device__ __forceinline__ static double term_of(const Pass &ps, const Row &r, int i, int mode) {
    const double eta = eta_of(ps, r, i, mode);
#if defined(AMWG_X_GLM_NOMATH)      // experiment (wrong results): the loads and the linear predictor only
    return eta + r.v[7] + r.v[8];
#endif
    double lam;
    const double lg = exp_log_v8(eta, lam, ps.K);
    return lg * r.v[7] - lam - r.v[8];
  }
  template <bool FAST>
  __device__ __forceinline__ static double term(const Pass &ps, int i) {
    Row r;
    load_row(ps, i, r);
    return term_of(ps, r, i, 1);
  }
  // The pass over the data with the NEXT observation's nine values (seven columns, the count, lfactorial) requested before the current
  // one's ~130 instructions start: the compiler's own schedule of the plain loop waits for each value right where it is used, i.e. a
  // whole L2 round trip per observation with nothing but the SIMD's other wave to cover it (round 2: 31 % of wave cycles in s_waitcnt).
  // Same terms, added in the same order.
  static constexpr bool kOwnPass = true;
#if defined(AMWG_X_GLM_WAVES)
  static constexpr int kMinWavesPerSimd = AMWG_X_GLM_WAVES;
#else
  static constexpr int kMinWavesPerSimd = 2;      // at most 256 VGPRs
#endif
  // Round 3, second half: 138 -> ~105 vector instructions per observation, all of them bookkeeping around the 86 fp64 operations the
  // expression needs:
  //  * exp and log fused (amwg_math.h exp_log_v8): log's split of its argument, its (double)k and both k*ln2 products are what exp just
  //    formed; exp's three-way choice of k is one formula;
  //  * the change point: every lane's FIRST row with i >= cp is known before the loop, so whole rounds (all lanes of the wave on one
  //    side) are told apart on the scalar unit -- `mode` 0: nobody adds b[7]; 2: everybody does; 1: the one or two rounds in between
  //    compare (integer compare, see begin());
  //  * the loop never asks for a row past a lane's last one (the clamp cost a compare, a select and two moves per row): the pipelined
  //    loop stops early, the tail is predicated;
  //  * constants of exp in vector registers: three of the nine base pointers had been spilled for them and were read back with
  //    v_readlane for every row;
  //  * eta starts at the first product instead of 0 + product (they differ for a product of -0 only, and a sum that is -0 instead of
  //    +0 in the end has the same exp);
  //  * two observations side by side through exp / log (pair_finish: one basic block, the two dependent chains interleave), the next two
  //    rows requested as soon as the linear predictors have consumed the current ones (same registers).
  // What it bought is less than the instruction count says (4.62 -> 5.0e6 updates/s): see DESIGN.md section 4 -- the pass sits on two
  // ceilings at once, the VALU issue rate (an fp64 instruction every ~4.45 cycles with two waves per SIMD, v_rcp_f64 16) and the vector
  // memory pipeline (72 B per observation and chain through L1: with exp and log REMOVED it runs only 6 % faster).  A workgroup-shared
  // LDS tiling of the data (a quarter of the L1 traffic) was built and measured 4 % SLOWER than this loop on the same box, and dropped.
  template <int G>
  __device__ __forceinline__ static double pass(const Pass &ps, int n_obs, int sub, double acc) {
    const int n_full = n_obs / G, rem = n_obs % G;
    const int n_mine = n_full + (sub < rem ? 1 : 0);           // this lane's observations: sub, sub + G, ...
    // first own round k with k*G + sub >= icp; over the wave: nobody adds before k_some, everybody from k_all on
    const int first = ps.icp <= sub ? 0 : (ps.icp - sub + (G - 1)) / G;
    // Only a chain that IS the wave (G >= 64) can agree on these over the wave: with several chains per wavefront the lanes of a wave hold
    // different change points AND run this pass under different execution masks (a chain whose proposal fell outside its bounds skips the
    // evaluation), and a butterfly over a partly masked wave does not reach every lane -- the first active lane's maximum then missed other
    // chains' later thresholds, which were told "everybody adds b[7] from here on" rounds too early.  (Found in round 4 by running the same
    // seeded job with one and with 64 lanes per chain, tests/test_gpu_decision_parity.py: one chain in five differed after 1e4 steps; the
    // goldens' handful of chains never met the case.)  Fewer lanes per chain: every round compares.
    int k_some = 0, k_all = 0x7fffffff;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (G >= 64) {
      int lo = first, hi = first;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const int l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
      }
      k_some = __builtin_amdgcn_readfirstlane(lo);
      k_all = __builtin_amdgcn_readfirstlane(hi);
    }
#else
    k_some = first; k_all = first;
#endif
    auto mode_of = [&](int k) { return k < k_some ? 0 : (k >= k_all ? 2 : 1); };
    Row a, b;
    int k = 0;
    if (n_full >= 4) {
      load_row(ps, sub, a);
      load_row(ps, G + sub, b);
      for (; k + 3 < n_full; k += 2) {                          // rounds k + 2, k + 3 < n_full: every lane has those rows
        const double eta_a = eta_of(ps, a, k * G + sub, mode_of(k)), eta_b = eta_of(ps, b, (k + 1) * G + sub, mode_of(k + 1));
        const double ya = a.v[7], la = a.v[8], yb = b.v[7], lb = b.v[8];
        AMWG_STAGE_FENCE();
        load_row(ps, (k + 2) * G + sub, a);
        load_row(ps, (k + 3) * G + sub, b);
        AMWG_STAGE_FENCE();
        acc = pair_finish(ps, eta_a, eta_b, ya, la, yb, lb, acc);
        AMWG_STAGE_FENCE();
      }
    }
    for (; k < n_mine; ++k) {                                   // the last rounds (two of them were requested above already: re-read)
      load_row(ps, k * G + sub, a);
      acc += term_of(ps, a, k * G + sub, 1);
    }
    return acc;
  }
  // CERTIFIED DECISIONS (16 lanes per chain: four chains to a wavefront; amwg_kernel.h).  The reference's term is  log(lambda) y - lambda - lfactorial(y)  with
  // lambda = exp(eta) -- it takes the LOGARITHM of the exponential it has just formed (ld.pois(y, Math.exp(eta)), distributions.js:282-284), which V8 returns to
  // within an ulp of eta but not as eta: 36 of the term's 86 operations.  As real numbers log_post = prior + sum eta_i y_i - sum e^eta_i - sum lfactorial(y_i); the
  // pass below forms exactly that -- eta by one product and six fmas, e^eta by exp_bounded (17 operations, no reciprocal), two running sums; the third sum is a
  // constant of the data -- ~32 operations per observation, and hands it to the stepper with a bound eps on how far it and the expression's value (log_post
  // above, in the chain's lane order) can be apart.
  //   And it is the WAVEFRONT's pass: the expression's pass re-reads the 72 bytes of every observation for every chain (3.6 MB per evaluation through L1 / L2:
  //   19 TB/s over the chip, the ceiling that kept this kernel at 0.55 of its arithmetic roof for three rounds -- with exp and log removed it ran 6 % faster).
  //   Here the 64 lanes of a wavefront share out the OBSERVATIONS whichever chain they belong to, the four chains' coefficients travel in scalar registers, and
  //   a row, once in registers, is evaluated for all four chains: a quarter of the memory traffic, and the kernel is bound by its arithmetic.  Four partial
  //   results per lane and one butterfly over the wavefront per chain leave each chain's total in its own lanes.
  // With u = 2^-53, H >= max_i |eta_i| (sum_k |b_k| max_i |x_ik| + |b_7|, from the column maxima the host keeps), Y = sum y_i, F = sum lfactorial(y_i), L = sum lambda_i:
  //   log(exp_v8(eta)) vs eta: 2 u (1 + H) 1.01 per unit of y; exp_v8 vs e^eta: 2 u L; the term's three roundings: 4 u (H Y + L + F); eta's 13 roundings against
  //   the 7 here: 22 u H (Y + L) 1.05; exp_bounded: 2^-46 L = 128 u L; the per-lane sums of n_l terms and the butterflies, on both sides: 2 (n_l + 8) u W; the final
  //   combination 4 u W -- W = |prior| + (1 + H) Y + L + F bounds every sum of magnitudes above, n_l = n / 16 + 1 (the expression's lanes hold the longer sums).
  //   eps = u W (2 n_l + 23 H + 200) 1.25  against the expression in the chain's 16-lane order.
  // What the bound below is FOR is the reference's own order (round 5, last part; reference_order below: the expression this kernel evaluates when a uniform falls
  // inside the bound, and the value a launch leaves behind): one running sum over the nine prior terms and the n observations in order -- n + 9 additions of partial
  // sums below W' = W + 2 |lunif_cp| (lane 0's start value is the rounded sum of two prior terms) in place of the lanes' n_l + 8:
  //   eps = u W' (n + n / 32 + 48 + 23 H + 200) 1.25  (7e-7 at cfg5; rounds 5 and 6 up to the last day carried 2 (n + 16) here -- the reference's n + 9 additions counted
  //   for BOTH sides, where this pass's own share is the lanes' 2 (n / 64 + 9): twice the bound, twice as many updates that evaluate the expression, each of which holds a
  //   wavefront for a millisecond at n = 5e4).
  // A decision certified with it is the REFERENCE's (one lane per chain), not merely this geometry's.
  // H > 690 (an eta could leave the range in which exp and log are ordinary), a negative count (F = +inf) or any non-finite value make eps non-finite: the
  // stepper then evaluates the expression.
  static constexpr bool kCertified = true;
  static constexpr int kCertifiedLanes = 16;
  struct Approx { double value, eps; };
  template <int G, int BT>
  __device__ __forceinline__ static Approx log_post_approx(Cache &kc, const StateView &S, const ModelConsts &mc, const DataRef &d, const unsigned char *smem, int sub) {
    static_assert(G == 16, "the certified pass of the Poisson family runs four chains to a wavefront");
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int CW = 64 / G;
    const int lane = (int)(threadIdx.x & 63u);
    const Pass ps = begin<G>(S, mc, d, smem, kc);      // this lane's chain: b[0..7], icp, the column pointers
    const double start = prior_split<G>(S, mc, d, sub, (sub == 0) ? prior(S, mc, d, kc) : 0.0, kc);
    // (what the bound needs of this lane's own chain, taken before the pass: its coefficients need not stay in registers through it)
    double H = __builtin_fabs(ps.b[7]);
#pragma unroll
    for (int q = 0; q < 7; ++q) H += __builtin_fabs(ps.b[q]) * mc.glm_xmax[q];
    const double P = butterfly<1, G>(start), Pabs = butterfly<1, G>(__builtin_fabs(start));
    const ExpTaylorRegs E = exp_taylor_regs();
    // the four chains' coefficients and thresholds, wave-uniform (v_readlane from the first lane of each chain: scalar registers for the whole pass)
    double bc[CW][8];
    int icp[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint64_t v = f64_bits(ps.b[k]);
        bc[c][k] = bits_f64(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), c * G) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, c * G));
      }
      icp[c] = __builtin_amdgcn_readlane(ps.icp, c * G);
    }
    double s1[CW], ls[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) { s1[c] = 0.0; ls[c] = 0.0; }
    auto load8 = [&](int i, double (&v)[8]) {
      const uint32_t off = (uint32_t)i * 8u;
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const double *>(ps.col[k] + off);      // seven columns and the count (col[7])
    };
    auto row = [&](const double (&v)[8], int i) {
      double eta[CW];
#pragma unroll
      for (int c = 0; c < CW; ++c) eta[c] = v[0] * bc[c][0];
#pragma unroll
      for (int k = 1; k < 7; ++k) {
#pragma unroll
        for (int c = 0; c < CW; ++c) eta[c] = __builtin_fma(v[k], bc[c][k], eta[c]);
      }
#pragma unroll
      for (int c = 0; c < CW; ++c) eta[c] = i >= icp[c] ? eta[c] + bc[c][7] : eta[c];
      double lam[CW];
#pragma unroll
      for (int c = 0; c < CW; ++c) lam[c] = exp_bounded(eta[c], E);
#pragma unroll
      for (int c = 0; c < CW; ++c) { s1[c] = __builtin_fma(eta[c], v[7], s1[c]); ls[c] += lam[c]; }
    };
    const int n_obs = d.n_obs, n_full = n_obs >> 6, rem = n_obs & 63;
    // (two row buffers, alternating: the round-5 loop copied the next row over the current one after every round -- 16 moves of the ~145 instructions a row costs;
    // the rows reach every lane's sums in the same order, so the value is the same double)
    double a[8], b[8];
    int k = 0;
    if (n_full >= 3) {
      load8(lane, a);
      for (; k + 2 < n_full; k += 2) {                           // rounds k + 1, k + 2 < n_full: every lane has those rows
        load8((k + 1) * 64 + lane, b);
        AMWG_STAGE_FENCE();
        row(a, k * 64 + lane);
        AMWG_STAGE_FENCE();
        load8((k + 2) * 64 + lane, a);
        AMWG_STAGE_FENCE();
        row(b, (k + 1) * 64 + lane);
        AMWG_STAGE_FENCE();
      }
      row(a, k * 64 + lane);                                     // (round k < n_full, loaded above)
      ++k;
    }
    for (; k < n_full + (lane < rem ? 1 : 0); ++k) {
      load8(k * 64 + lane, a);
      row(a, k * 64 + lane);
    }
    // every chain's totals over the wavefront; a lane keeps its own chain's
    const int mine = lane / G;
    double tot = 0.0, L = 0.0;
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const double t = butterfly<1, 64>(s1[c] - ls[c]), l = butterfly<1, 64>(ls[c]);
      tot = mine == c ? t : tot;
      L = mine == c ? l : L;
    }
    const double W = Pabs + 2.0 * __builtin_fabs(mc.lunif_cp) + (1.0 + H) * mc.glm_sum_y + L + mc.glm_sum_lf;
    const double eps = (H <= 690.0) ? W * ((double)n_obs + (double)(n_obs >> 5) + 48.0 + 23.0 * H + 200.0) * 1.25 * 0x1p-53 : __builtin_inf();
    return Approx{(P + tot) - mc.glm_sum_lf, eps};
#else
    return Approx{0.0, __builtin_inf()};
#endif
  }
  // THE REFERENCE'S ORDER at G lanes per chain: lp = 0; for k: lp += prior(b_k); lp += prior(cp); for i: lp += term_i (the closure of tests/model_spec.py, called by
  // mcmc.js:524-526) -- the chain's lanes compute the terms of a round of G observations side by side (term_of: the operations of the one-lane kernel on the same
  // values) and ONE running sum takes them in the order i = G k + lane (a ds_bpermute broadcast per term: ~1 ms for 5e4 observations, which is why it is not the
  // pass).  Runs under the chain's own execution mask: the lanes it reads are its own.
  static constexpr bool kReferenceOrder = true;
  template <int G>
  __device__ inline __attribute__((noinline)) static double reference_order_sum(const double *state, const double *x, const double *y, const double *lfact, int n_obs, int sub,
                                                                                double m0, double c0, double den0, double y0h, double y0l, int den0_ok, double cp_upper, double lunif_cp) {
    double acc = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
    Pass ps;
#pragma unroll
    for (int k = 0; k < 8; ++k) ps.b[k] = state[k];
    ps.cp = state[8];
    ps.icp = !(ps.cp < 536870912.0) ? 536870912 : (ps.cp <= 0.0 ? 0 : (int)__builtin_ceil(ps.cp));      // (as begin())
    ps.K = exp_log_regs();
#pragma unroll
    for (int k = 0; k < 7; ++k) ps.col[k] = reinterpret_cast<const char *>(x + (size_t)k * (size_t)n_obs);
    ps.col[7] = reinterpret_cast<const char *>(y);
    ps.col[8] = reinterpret_cast<const char *>(lfact);
    for (int k = 0; k < 8; ++k) acc += norm_const_sd(ps.b[k], m0, c0, den0, y0h, y0l, den0_ok);
    acc += (ps.cp < 0 || ps.cp > cp_upper) ? -kInf : lunif_cp;
    const int base = (int)(threadIdx.x & 63u) & ~(G - 1);
    for (int k0 = 0; k0 < n_obs; k0 += G) {
      const int cnt = n_obs - k0 < G ? n_obs - k0 : G, i = k0 + sub;
      Row r;
      load_row(ps, sub < cnt ? i : k0, r);
      const double term = term_of(ps, r, i, 1);
      for (int l = 0; l < cnt; ++l) acc += __shfl(term, base + l, 64);
    }
#endif
    return acc;
  }
  template <int G>
  __device__ __forceinline__ static double reference_order(Cache &, const StateView &S, const ModelConsts &mc, const DataRef &d, const unsigned char *, int sub) {
    return reference_order_sum<G>(S.base, d.x, d.y, d.lfact, d.n_obs, sub, mc.m0, mc.c0, mc.den0, mc.y0_hi, mc.y0_lo, mc.den0_ok, mc.cp_upper, mc.lunif_cp);
  }
  // two observations through exp / log side by side (one basic block: the two dependent chains interleave), added in order
  __device__ __forceinline__ static double pair_finish(const Pass &ps, double eta_a, double eta_b, double ya, double la, double yb, double lb, double acc) {
    const double eta[2] = {eta_a, eta_b};
    double lam[2], lg[2];
    bool rare[2];
    exp_log_v8_open<2>(eta, lam, lg, rare, ps.K);
    if (__builtin_expect(rare[0] | rare[1], 0)) {
      if (rare[0]) { lam[0] = exp_v8_cold(eta[0]); lg[0] = log_v8_cold(lam[0]); }
      if (rare[1]) { lam[1] = exp_v8_cold(eta[1]); lg[1] = log_v8_cold(lam[1]); }
    }
    acc += lg[0] * ya - lam[0] - la;
    acc += lg[1] * yb - lam[1] - lb;
    return acc;
  }
};

}  // namespace amwg

#include "amwg_gl.h"      // the group-local kernel of the hierarchical family (HierGlModel)
