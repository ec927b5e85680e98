// amwg_gl.h -- the GROUP-LOCAL evaluation of the hierarchical Normal family (opt-in: amwg_options::group_local), as its own kernel.
//
//     y_i ~ norm(theta[g_i], sigma);  theta_k ~ norm(mu, tau);  mu ~ norm(m0, s0);  sigma ~ unif(a, b)          SURVEY.md section 8(d) cfg4
//
// mcmc.js:524-526 evaluates the whole log_post twice per update.  Of this model's Gn + 2 updates per step, Gn touch one group (its
// observations and one prior term) and one touches no observation at all.  Here a chain lives on one wavefront and
//   * a proposal for mu re-evaluates the prior terms only, one for sigma makes ONE pass over the data;
//   * the Gn proposals for theta of a step are evaluated TOGETHER in one pass -- every lane with the proposed mean of the one group it
//     serves -- and each is decided on its local difference (pt' - pt) + (L' - L).
// Two passes per step instead of Gn + 2.  NOT the reference's operation schedule: the order of additions it follows is restated, and tested
// bit for bit, in oracle/amwg_oracle.c (gl_*); its accept decisions equal the reference's unless the accept uniform falls inside the
// ~1e-12-relative sliver between the two orders' exp(delta) (counted: tests/test_gpu_decision_parity.py, tools/flip_rate.py).
//
// Round 4: any labels g_i in [0, Gn), any Gn <= 64 (round 3: g_i = i mod Gn, Gn a power of two).  The host deals the 64 lanes to the groups
// (gl_layout, amwg_core.hip; restated in the oracle): group k gets an ALIGNED BLOCK of L_k = 2^j lanes -- one lane each to begin with, then the
// group with the most observations per lane is doubled while lanes are left -- blocks placed in order of decreasing size (ties: group index).
// Lane m of block k takes the group's observations number m, m + L_k, ... (in index order).  The data is staged LANE-MAJOR in LDS,
// tile[r * 64 + lane] = the r-th observation of that lane (conflict-free: the 64 lanes read 64 consecutive doubles), rounds beyond a lane's own
// count are masked.  With that
//     T_j   = the data-only sum of lane j (0 + term + term + ..., its group's mean)
//     L_k   = the butterfly of T over the lanes of block k (offsets 1, 2, .. L_k / 2: the block is aligned, partners stay inside)
//     pt_k  = ld.norm(theta_k, mu, tau),  pm = ld.norm(mu, m0, s0) + ld.unif(sigma, a, b)
//     log_post_GL = the butterfly over all 64 lanes of V_j:  V_0 = (pm + pt) + T_0;  V_j = pt + T_j on the first lane of a block;  V_j = T_j
// The sequential semantics of the reference's sweep is kept through the UNIFORM STREAM: which (u, v) pair of the stream survives rnorm's
// rejection test (mcmc.js:44-53) is a property of the stream alone, so the acceptance flags of the pairs of a 256-uniform window are
// computed lane-parallel, a scalar loop walks the Gn updates in their shuffled order -- first accepted pair at or after the position (a
// find-first-set on the flag words), then one accept uniform if the proposal is inside its bounds -- and every lane reads the three uniforms
// of its group's update from the window (kept in LDS, 2 KB per wavefront).
//
// What round 4 changed in the sweep, measured on round 3's profile (141 VALU + 69 SALU per update, 1e7 LDS bank conflicts per launch):
//   * the per-step Fisher-Yates shuffle of the Gn components (mcmc.js:248-252) consumed a quarter of the non-pass instructions as 31 dependent
//     rounds of { draw, 2 x v_readlane, selects }.  Its Gn - 1 uniforms are known positions of the stream: every lane fetches and scales ITS one
//     (j_i = floor(u_i * (i + 1))) at once, and only the 31 transpositions stay sequential -- applied to each lane's POSITION in the order (two
//     compares + two selects per transposition);
//   * the window persists across sweeps (a Philox block per lane buys 128 uniforms; round 3 recomputed the second half in every sweep) and
//     lives in LDS, so a lane's three uniforms are three ds_read_b64 instead of 24 ds_bpermute + selects (the bank conflicts);
//   * the resolution loop writes update t's stream position into lane t (one v_writelane) instead of comparing and selecting in every lane;
//   * its own kernel: the ordinary stepper's code, registers and scalar constants are not carried through the sweep.
#pragma once
// (included at the end of amwg_models.h: NormCache, norm_const_sd, HierNormalModel::prior_mu_sigma_cold, the staged pass)

namespace amwg {

// one entry per lane of the wavefront (amwg_core.hip gl_layout; d.arr[0])
struct GlLane {
  int32_t cnt;      // observations of this lane
  int8_t grp;       // its group = its component of theta (-1: no group, an idle lane)
  int8_t blk;       // lanes of its block (a power of two)
  int8_t first;     // 1 = first lane of its block: holds the group's prior term in log_post_GL and does the component's bookkeeping
  int8_t first_of;  // lane c (c < Gn): the first lane of component c's block
};

struct HierGlModel {
  static constexpr bool kUser = false, kHasFast = false, kOneLanePass = false, kSplitPrior = false;
  static constexpr int kDerived = 0;
  static constexpr bool kHasBinary = false;
  static constexpr int kMaxThreads = 1024;
  static constexpr int kUnroll = 8;
  static constexpr bool kGroupLocalKernel = true;
  // LDS data region of a workgroup: the lane-major tile | the lane table | one 256-uniform window per wavefront
  static constexpr size_t kTableBytes = 64 * sizeof(GlLane), kWindowBytes = 256 * 8;
  __host__ __device__ static size_t gl_lds_bytes(int rounds, int waves) { return (size_t)rounds * 64 * 8 + kTableBytes + (size_t)waves * kWindowBytes; }
  __host__ __device__ static size_t lds_bytes(int, int, int) { return 0; }      // (the group-local kernel sizes its data region with gl_lds_bytes)
  // DataRef of a group-local sampler: x = the tile [rounds][64], arr[0] = GlLane[64], pad = rounds (the largest lane count), K = the smallest
  __device__ static void stage(unsigned char *smem, const DataRef &d, int tid, int nt, int) {
    double *dst = reinterpret_cast<double *>(smem);
    const int n = d.pad * 64;
    for (int i = tid; i < n; i += nt) dst[i] = d.x[i];
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem + (size_t)n * 8);
    const uint32_t *src = static_cast<const uint32_t *>(d.arr[0]);
    for (int i = tid; i < (int)(kTableBytes / 4); i += nt) tab[i] = src[i];
  }

  using Stream = WindowStream;      // (amwg_window.h)

  // ---- what every lane keeps of ITS group between evaluations
  struct Lane {
    NormCache n;                       // sd-dependent constants of the pass (of the sd last evaluated)
    double mu, sigma, th;              // th: the mean of this lane's group (0 on an idle lane)
    double pm, pt, T, Ls;              // committed pieces of log_post_GL: prior(mu, sigma) | prior term of th | this lane's data sum | its block's
    double pa, pb;                     // pm = pa + pb: ld.norm(mu, m0, s0) and ld.unif(sigma, a, b) -- a proposal for one of the two leaves the other term as it is
    double pm_t, pt_t, T_t, pa_t, pb_t;   // the same of the proposal being evaluated
    double c1, den1, y1h, y1l;         // constants of theta's prior, in vector registers
    int den1_ok;
    int grp, blk, cnt;
    bool first;
  };
  // the closure's `lp = 0; lp += ld.norm(mu, m0, s0); lp += ld.unif(sigma, a, b)` is (0 + pa) + pb = pa + pb, bit for bit (0 + x = x for every x but -0,
  // which c0 - q never is).  The hyper-parameters come straight from the kernel's argument block (scalar registers): no call, no load to wait for
  __device__ __forceinline__ static double prior_mu(double mu, const ModelConsts &mc) { return norm_const_sd(mu, mc.m0, mc.c0, mc.den0, mc.y0_hi, mc.y0_lo, mc.den0_ok); }
  __device__ __forceinline__ static double prior_sigma(double sigma, const ModelConsts &mc) { return (sigma < mc.ua || sigma > mc.ub) ? -kInf : mc.lunif; }
  __device__ __forceinline__ static double prior_theta(const Lane &k, double theta, double mu) {
    return norm_const_sd(theta, mu, k.c1, k.den1, k.y1h, k.y1l, k.den1_ok);
  }
  // the value this lane contributes to log_post_GL
  __device__ __forceinline__ static double lane_value(const Lane &k, int lane, double pm, double pt, double T) {
    return lane == 0 ? (pm + pt) + T : (k.first ? pt + T : T);
  }
  __device__ __forceinline__ static double total(const Lane &k, int lane, double pm, double pt, double T) {
    return butterfly<1, 64>(lane_value(k, lane, pm, pt, T));
  }
  // the sum of T over the lanes of this lane's block: the stages of the butterfly below the block size (the block is aligned: partners are
  // inside it, and a lane of a smaller block simply keeps its value)
  __device__ __forceinline__ static double block_sum(double T, int blk) {
    double s;
    s = xor_sum<1>(T); T = blk > 1 ? s : T;
    s = xor_sum<2>(T); T = blk > 2 ? s : T;
    s = xor_sum<4>(T); T = blk > 4 ? s : T;
    s = xor_sum<8>(T); T = blk > 8 ? s : T;
    s = xor_sum<16>(T); T = blk > 16 ? s : T;
    s = xor_sum<32>(T); T = blk > 32 ? s : T;
    return T;
  }
  // T of this lane for the given mean of its group, with the sd the NormCache holds: the first n_min rounds every lane has (hand-scheduled
  // pass), then the rounds only some lanes have
  __device__ inline __attribute__((noinline)) static double pass_slow(const double *tile, double mean, double c, double den, int n_max, int cnt, int lane) {
    double acc = 0.0;
    for (int r = 0; r < n_max; ++r) { const double t = tile[r * 64 + lane] - mean; const double term = c - (t * t) / den; acc = r < cnt ? acc + term : acc; }
    return acc;
  }
  // Same operations on the same values in the same order as the plain loop `acc = 0; for r < cnt: acc += c - (x_r - mean)^2 / den`: the
  // first n_min rounds (every lane that serves a group has them) through the hand-scheduled pass, the rest in masked blocks of U.  An idle
  // lane computes on the tile's padding and returns 0.
  template <int U>
  __device__ __forceinline__ static double pass(const Lane &k, const ModelConsts &mc, const DataRef &d, const unsigned char *smem, int lane, double mean) {
    const double *tile = reinterpret_cast<const double *>(smem);
    const int n_min = fresh_uniform(d.K), n_max = fresh_uniform(d.pad);
    const bool mine = mean == 0 || mid_range(__builtin_fabs(mean));
    const bool ok = !mc.exact_division && mc.data_mid_range && k.n.den_ok && __ballot(mine) == ~0ull;
    if (!ok) return pass_slow(tile, mean, k.n.c, k.n.den, n_max, k.cnt, lane);
    double acc = norm_pass_staged<64, U, false>(tile, nullptr, StateView{nullptr}, mean, k.n.c, k.n.den, k.n.y, n_min * 64, lane, 0.0);
    if (n_max - n_min <= 2) {      // a balanced design: one or two ragged rounds, term by term (a masked block would spend U slots on them)
      double xv[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) xv[q] = tile[(n_min + q < n_max ? n_min + q : n_max - 1) * 64 + lane];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (n_min + q < n_max) {      // uniform
          const double t = xv[q] - mean;
          const double term = k.n.c - div_by_invariant(t * t, k.n.den, k.n.y);
          acc = n_min + q < k.cnt ? acc + term : acc;
        }
      }
    } else
    for (int r0 = n_min; r0 < n_max; r0 += U) {
      NormBlock<U> xt, qt;
      double mt[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = r0 + u < n_max ? r0 + u : n_max - 1;      // (rows past the end: any valid address, masked below)
        xt.v[u] = tile[r * 64 + lane];
        mt[u] = mean;
      }
      AMWG_STAGE_FENCE();
      norm_block_stages<U, false>(xt, mt, qt, qt, acc, k.n.c, k.n.den, k.n.y);
#pragma unroll
      for (int u = 0; u < U; ++u) acc = r0 + u < k.cnt ? acc + qt.v[u] : acc;
    }
    return k.cnt > 0 ? acc : 0.0;
  }
  // everything from the state as it stands (launch start; equals what the previous launch ended with, bit for bit) -> log_post_GL
  template <int U>
  __device__ __forceinline__ static double refresh(Lane &k, const StateView &S, const ModelConsts &mc, const DataRef &d, const unsigned char *smem, int lane) {
    const GlLane *tab = reinterpret_cast<const GlLane *>(smem + (size_t)d.pad * 64 * 8);
    const GlLane me = tab[lane];
    k.grp = me.grp; k.blk = me.blk; k.cnt = me.cnt; k.first = me.first != 0;
    k.mu = S(d.G);
    k.sigma = S(d.G + 1);
    k.th = k.grp >= 0 ? S(k.grp) : 0.0;
    k.c1 = mc.c1; k.den1 = mc.den1; k.y1h = mc.y1_hi; k.y1l = mc.y1_lo; k.den1_ok = mc.den1_ok;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(k.c1), "+v"(k.den1), "+v"(k.y1h), "+v"(k.y1l), "+v"(k.den1_ok));      // (vector registers from here on)
#endif
    k.n = norm_cache_init();
    norm_cache_update<true>(k.n, k.sigma, mc.neg_half_log_2pi);
    k.pa = prior_mu(k.mu, mc);
    k.pb = prior_sigma(k.sigma, mc);
    k.pm = k.pa + k.pb;
    k.pt = k.grp >= 0 ? prior_theta(k, k.th, k.mu) : 0.0;
    k.T = pass<U>(k, mc, d, smem, lane, k.th);
    k.Ls = block_sum(k.T, k.blk);
    k.pm_t = k.pm; k.pt_t = k.pt; k.T_t = k.T; k.pa_t = k.pa; k.pb_t = k.pb;
    return total(k, lane, k.pm, k.pt, k.T);
  }
  // a proposal v for mu (is_mu) or sigma: log_post_GL of the proposed state, its pieces kept as tentative
  template <int U>
  __device__ __forceinline__ static double eval_scalar(Lane &k, bool is_mu, double v, const ModelConsts &mc, const DataRef &d, const unsigned char *smem, int lane) {
    if (is_mu) {
      k.pa_t = prior_mu(v, mc);
      k.pb_t = k.pb;
      k.pt_t = k.grp >= 0 ? prior_theta(k, k.th, v) : 0.0;
      k.T_t = k.T;
    } else {
      k.pa_t = k.pa;
      k.pb_t = prior_sigma(v, mc);
      k.pt_t = k.pt;
      norm_cache_update<true>(k.n, v, mc.neg_half_log_2pi);
#if defined(AMWG_X_GLCUT) && AMWG_X_GLCUT == 5
      k.T_t = k.T;
#else
      k.T_t = pass<U>(k, mc, d, smem, lane, k.th);
#endif
    }
    k.pm_t = k.pa_t + k.pb_t;
    return total(k, lane, k.pm_t, k.pt_t, k.T_t);
  }
  __device__ __forceinline__ static void commit_scalar(Lane &k, bool is_mu, double v) {
    k.pm = k.pm_t; k.pt = k.pt_t; k.pa = k.pa_t; k.pb = k.pb_t;
    if (!is_mu) { k.T = k.T_t; k.Ls = block_sum(k.T_t, k.blk); }      // (wave-uniform branch: every lane of the chain decided alike)
    k.mu = is_mu ? v : k.mu;
    k.sigma = is_mu ? k.sigma : v;
  }
  // the sweep over theta: every lane evaluates the proposal of its own group (`eval`: that proposal is evaluated in this round, `prop` its
  // value -- the same on all lanes of the block) -> the local difference its accept test uses; L' is left in Ls_t
  template <int U>
  __device__ __forceinline__ static double sweep_eval(Lane &k, bool eval, double prop, double &Ls_t, const ModelConsts &mc, const DataRef &d, const unsigned char *smem, int lane) {
    norm_cache_update<true>(k.n, k.sigma, mc.neg_half_log_2pi);       // (a rejected sigma proposal leaves the cache at the proposed sd)
    const double mean = eval ? prop : k.th;
    k.pt_t = eval ? prior_theta(k, prop, k.mu) : k.pt;
#if defined(AMWG_X_GLCUT) && AMWG_X_GLCUT == 3
    k.T_t = k.T + mean * 1e-300;
#else
    k.T_t = pass<U>(k, mc, d, smem, lane, mean);
#endif
    Ls_t = block_sum(k.T_t, k.blk);
    return (k.pt_t - k.pt) + (Ls_t - k.Ls);
  }
  __device__ __forceinline__ static void sweep_commit(Lane &k, bool accepted, double prop, double Ls_t) {
    k.th = accepted ? prop : k.th;
    k.pt = accepted ? k.pt_t : k.pt;
    k.T = accepted ? k.T_t : k.T;
    k.Ls = accepted ? Ls_t : k.Ls;
  }
};

}  // namespace amwg
