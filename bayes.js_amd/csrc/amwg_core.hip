// amwg_core.hip -- host side of libamwg.so: the C ABI of include/amwg.h over the fused
// gfx950 step kernel (amwg_kernel.h).  No torch, no oracle, no CPU fallback: every entry
// point that computes runs on the HIP device or fails with AMWG_EHIP.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/amwg.h"
#include "amwg_build_id.h"
#if defined(AMWG_SELFTEST) || defined(AMWG_AUDIT) || defined(AMWG_X_PHASES)
#include "../../include/amwg_selftest.h"
#endif
#if defined(AMWG_SELFTEST)
#include "amwg_eval.h"      // device evaluation of the arithmetic building blocks: libamwg_selftest.so only
#endif
#include "amwg_kernel.h"
#include "amwg_models.h"
#include "amwg_sampler.h"

using namespace amwg;

// The kernel headers as text (amwg_rtc_headers.c, .incbin): hiprtc compiles a translated closure
// together with the very same step kernel source the built-in models are compiled from.
extern "C" {
extern const char amwg_hdr_stdint[], amwg_hdr_types[], amwg_hdr_math[], amwg_hdr_div[], amwg_hdr_ld[], amwg_hdr_philox[],
    amwg_hdr_kernel[], amwg_hdr_user[], amwg_hdr_twoval[], amwg_hdr_kval[], amwg_hdr_trig[], amwg_hdr_pass[], amwg_hdr_rows[], amwg_hdr_window[], amwg_hdr_ptail[];
}

// the step kernels of the built-in families, one translation unit each (amwg_kernels.hip): kernel for (lanes per chain, workgroup size)
step_kernel_t amwg_kernels_normal(int lanes, int block);
step_kernel_t amwg_kernels_beta_bern(int lanes, int block);
step_kernel_t amwg_kernels_hier_normal(int lanes, int block);
step_kernel_t amwg_kernels_pois_glm(int lanes, int block);
step_kernel_t amwg_kernel_hier_gl(int block);      // the group-local kernel of the hierarchical family (amwg_gl.h)
step_kernel_t amwg_kernel_hier_sweep(int block);   // the hierarchical family's kernel with the sweep prefetch (row layout, 64 lanes per chain)
// ... and the kernels that decide from certified values (amwg_kernel.h kCert), where a family has one for this lane count (nullptr: none)
step_kernel_t amwg_kernels_cert_normal(int lanes, int block);
step_kernel_t amwg_kernels_cert_beta_bern(int lanes, int block);
step_kernel_t amwg_kernels_cert_hier_normal(int lanes, int block);
step_kernel_t amwg_kernels_cert_pois_glm(int lanes, int block);

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) return fail(AMWG_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)


step_kernel_t pick_kernel(int model, int G, int block) {
  switch (model) {
    case AMWG_MODEL_NORMAL: return amwg_kernels_normal(G, block);
    case AMWG_MODEL_BETA_BERN: return amwg_kernels_beta_bern(G, block);
    case AMWG_MODEL_HIER_NORMAL: return amwg_kernels_hier_normal(G, block);
    case AMWG_MODEL_POIS_GLM: return amwg_kernels_pois_glm(G, block);
  }
  return nullptr;
}
step_kernel_t pick_certified_kernel(int model, int G, int block) {
  switch (model) {
    case AMWG_MODEL_NORMAL: return amwg_kernels_cert_normal(G, block);
    case AMWG_MODEL_BETA_BERN: return amwg_kernels_cert_beta_bern(G, block);
    case AMWG_MODEL_HIER_NORMAL: return amwg_kernels_cert_hier_normal(G, block);
    case AMWG_MODEL_POIS_GLM: return amwg_kernels_cert_pois_glm(G, block);
  }
  return nullptr;
}

size_t model_lds_bytes(int model, int n_obs, int groups, int lanes) {
  switch (model) {
    case AMWG_MODEL_NORMAL: return NormalModel::lds_bytes(n_obs, groups, lanes);
    case AMWG_MODEL_BETA_BERN: return BetaBernModel::lds_bytes(n_obs, groups, lanes);
    case AMWG_MODEL_HIER_NORMAL: return HierNormalModel::lds_bytes(n_obs, groups, lanes);
    case AMWG_MODEL_POIS_GLM: return PoisGlmModel::lds_bytes(n_obs, groups, lanes);
  }
  return 0;
}

int model_max_threads(int model) {
  switch (model) {
    case AMWG_MODEL_NORMAL: return NormalModel::kMaxThreads;
    case AMWG_MODEL_BETA_BERN: return BetaBernModel::kMaxThreads;
    case AMWG_MODEL_HIER_NORMAL: return HierNormalModel::kMaxThreads;
    case AMWG_MODEL_POIS_GLM: return PoisGlmModel::kMaxThreads;
  }
  return 64;
}

bool value_mid_range(double v) { return v == 0.0 || mid_range(std::fabs(v)); }

// scratch device buffer of the helper entry points: freed on every exit path
struct DevBuf {
  void *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
  template <class T> T *as() const { return static_cast<T *>(p); }
};
struct EventPair {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ~EventPair() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
};

}  // namespace


namespace {

template <class T>
int dev_alloc(amwg_sampler *s, T **p, size_t n) {
  void *q = nullptr;
  HIP_TRY(hipMalloc(&q, n * sizeof(T) ? n * sizeof(T) : 1));
  s->dev_allocs.push_back(q);
  *p = static_cast<T *>(q);
  return AMWG_OK;
}

// Group-local evaluation (amwg_gl.h): deals the 64 lanes of a chain's wavefront to the groups and lays the observations out lane-major.
//   * lanes: every group one lane to begin with (Gn <= 64); then, while lanes are left, the group with the most observations per lane
//     (ceil(n_k / L_k); ties: the smaller index) has its lane count doubled -- it stops when that group cannot be doubled any more, since
//     doubling others would not shorten the longest lane;
//   * placement: blocks in order of decreasing size (ties: group index), so that every block of 2^j lanes starts at a multiple of 2^j;
//   * observations: lane m of block k takes the group's observations (in index order) number m, m + L_k, m + 2 L_k, ...;
//   * tile[r * 64 + lane] = the r-th observation of that lane, rows padded with 0 up to the longest lane.
// Restated in oracle/amwg_oracle.c (gl_layout) -- the order of additions of the group-local mode follows from it.
struct GlLayoutHost {
  std::vector<double> tile;
  GlLane lane[64];
  int rounds = 0, n_min = 0;
};
int gl_layout(const double *y, const int32_t *g, int N, int Gn, GlLayoutHost *out) {
  if (Gn < 1 || Gn > 64) return fail(AMWG_EINVAL, "group_local: 1 to 64 groups (a chain runs on one wavefront, a lane serves one group), got %d", Gn);
  std::vector<int> n(Gn, 0), L(Gn, 1), first(Gn, 0);
  for (int i = 0; i < N; ++i) {
    if (g[i] < 0 || g[i] >= Gn) return fail(AMWG_EINVAL, "group_local: g[%d] = %d outside 0..%d", i, g[i], Gn - 1);
    n[g[i]]++;
  }
  int total = Gn;
  for (;;) {
    int best = 0;
    long load_best = -1;
    for (int k = 0; k < Gn; ++k) { const long load = (n[k] + L[k] - 1) / L[k]; if (load > load_best) { load_best = load; best = k; } }
    if (load_best <= 1 || total + L[best] > 64) break;
    total += L[best];
    L[best] *= 2;
  }
  std::vector<int> order(Gn);
  for (int k = 0; k < Gn; ++k) order[k] = k;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return L[a] > L[b]; });
  for (int j = 0; j < 64; ++j) out->lane[j] = GlLane{0, -1, 1, 0, 0};
  int at = 0;
  for (int k : order) {
    first[k] = at;
    for (int m = 0; m < L[k]; ++m) {
      GlLane &q = out->lane[at + m];
      q.grp = (int8_t)k; q.blk = (int8_t)L[k]; q.first = m == 0 ? 1 : 0;
      q.cnt = (n[k] - m + L[k] - 1) / L[k];
      if (q.cnt < 0) q.cnt = 0;
    }
    at += L[k];
  }
  for (int c = 0; c < Gn; ++c) out->lane[c].first_of = (int8_t)first[c];
  int rounds = 0, n_min = -1;
  for (int j = 0; j < 64; ++j) {
    if (out->lane[j].cnt > rounds) rounds = out->lane[j].cnt;
    if (out->lane[j].cnt > 0 && (n_min < 0 || out->lane[j].cnt < n_min)) n_min = out->lane[j].cnt;
  }
  if (rounds < 1) rounds = 1;
  if (n_min < 0) n_min = 0;
  out->rounds = rounds;
  out->n_min = n_min;
  out->tile.assign((size_t)rounds * 64, 0.0);
  std::vector<int> seen(Gn, 0);
  for (int i = 0; i < N; ++i) {
    const int k = g[i], m = seen[k] % L[k], r = seen[k] / L[k];
    out->tile[(size_t)r * 64 + first[k] + m] = y[i];
    seen[k]++;
  }
  return AMWG_OK;
}

// Geometry.  For every lanes-per-chain G take the largest workgroup that still gives every CU a workgroup (more waves
// share one LDS copy of the data) and price it with a two-term model of one parameter update:
//     cost(G) = rounds * [ S(G) * max(w_res, 1.8) + (W / G) * max(w_res, 1.15) * (1 + 0.3 / w_res) ]
// S(G) = the replicated stepper (Philox, proposal, exp, accept, adaptation), W/G the wave's share of the log-likelihood
// work, w_res the number of waves a SIMD holds at once (limited by LDS and by the number of chains), rounds the number of
// such batches; the floors are the occupancies below which each part is latency- rather than issue-bound.  The cheapest
// G wins, ties go to the smaller G.  The choice depends only on the model, the data size and the chain count, so a
// given sampler configuration always gets the same G (the lane count fixes the summation order, hence the draws).
bool hier_rows_wanted(const amwg_sampler *s, int G);
// would this geometry run a kernel that decides from certified values (amwg_kernel.h kCert)?  The Normal family at one lane per chain, the Poisson family at 16,
// the hierarchical family's sweep kernel (64 lanes, the row layout in use); options.full_evaluation = 0, no exact_division, no group_local, not a closure
static bool certified_wanted(const amwg_sampler *s, int lanes, bool rows) {
  if (s->user || s->opt.full_evaluation != 0 || s->opt.exact_division || s->mc.group_local) return false;
  return (s->model == AMWG_MODEL_NORMAL && lanes == 1) || (s->model == AMWG_MODEL_POIS_GLM && lanes == 16) || (s->model == AMWG_MODEL_HIER_NORMAL && lanes == 64 && rows);
}
bool hier_rows_fit(const amwg_sampler *s, int bt, size_t max_lds);
// the Normal family at one lane per chain stages its observations in LDS only for the wavefront's certified pass (NormalModel::lds_bytes_of: DataRef::pad = 1)
static bool normal_tile_wanted(const amwg_sampler *s, int bt) { return !s->user && s->model == AMWG_MODEL_NORMAL && certified_wanted(s, 1, false) && bt <= 512 && !s->opt.sufficient_statistics; }
// ... a translated closure with a certified tail (amwg_user.h norm_tail_approx; read off the generated source by amwg_create_user): one lane per chain
// ... or a certified Poisson tail (amwg_ptail.h pois_tail_approx; kPoisTail of the generated source): 16 lanes per chain, four chains sharing every row they read
static bool user_cert_wanted(const amwg_sampler *s, int lanes) {
  if (!s->user || s->opt.full_evaluation != 0 || s->opt.exact_division || s->user_has_binary) return false;
  return (s->user_cert_tail_n > 0 && lanes == 1) || (s->user_pois_tail_n > 0 && lanes == 16);
}
// ... and a closure whose row plan the translator marked kRowCert: the sweep kernel decides from certified values, against the expression in the reference's order
static bool user_rows_cert_wanted(const amwg_sampler *s) { return s->user && s->user_rows_cert && s->opt.full_evaluation == 0 && !s->opt.exact_division && !s->user_has_binary; }
bool user_rows_wanted(const amwg_sampler *s, int G);
bool user_rows_fit(const amwg_sampler *s, int bt, size_t max_lds);

double model_work(const amwg_sampler *s, int G) {
  const double N = (double)s->d.n_obs;
  switch (s->model) {
    // (one lane per chain: accept tests are decided from the certified pass -- two operations per observation -- unless the caller asked for the expression in every update)
    case AMWG_MODEL_NORMAL: return (G == 1 && s->opt.full_evaluation == 0 && !s->opt.exact_division) ? (s->opt.sufficient_statistics ? 40.0 : 2.6 * N) : 9.0 * N;
    case AMWG_MODEL_BETA_BERN:   // one lane: exact fast-forward over ~log2(N) binades (or the scalar jump-table pass, one add per observation)
      return G == 1 ? (s->mc.exact_division ? 1.8 * N : 400.0 * (1.0 + std::log2(N + 2.0))) : 6.0 * N;
    case AMWG_MODEL_HIER_NORMAL: {
      // group labels that repeat with the lane stride: a lane reads its one mean once (constant-mean pass, 8.1 VALU and 1 LDS read per
      // observation); otherwise the gathered pass (9.4 VALU, 2.5 LDS reads: the LDS pipe, not the VALU, then sets the pace)
      int lg = 0;
      for (int g = G; g > 1; g >>= 1) ++lg;
      const bool periodic = ((s->hier_periodic_mask >> lg) & 1u) != 0;
      double w = (periodic ? 8.6 : 12.0) * N + 12.0 * s->d.G;
      // lane-local re-evaluation (row layout): of the G + 2 updates of a step only two make the full pass, the others re-form the sums of
      // the lanes of one group (~0.3 of a pass in time: a dependent chain on one lane)
      if (hier_rows_wanted(s, G) && hier_rows_fit(s, 256, (size_t)160 * 1024)) w *= (2.0 + 0.35 * s->d.G) / (2.0 + s->d.G);
      return w;
    }
    case AMWG_MODEL_POIS_GLM: return (G == 16 && s->opt.full_evaluation == 0 && !s->opt.exact_division) ? 36.0 * N : 90.0 * N;      // (16 lanes per chain: the certified pass, four chains sharing every row they read)
  }
  if (user_cert_wanted(s, G) && s->user_pois_tail_n > 0) {      // exp + log per observation (~70 of the term's operations) become exp_bounded's 19, and a row is read once for four chains
    const double n = (double)s->user_pois_tail_n, w = s->user_work > 0 ? s->user_work : 1e6;
    return (w - 70.0 * n > 0.4 * w) ? w - 70.0 * n : 0.4 * w;
  }
  if (user_cert_wanted(s, G)) {      // the tail loop's ~16 instructions per observation become the certified pass's 2.6
    const double w1 = s->user_work_one_lane > 0 ? s->user_work_one_lane : s->user_work, n = (double)s->user_cert_tail_n;
    return (w1 - 16.0 * n > 0 ? w1 - 16.0 * n : 0.0) + 2.6 * n;
  }
  if (G == 1 && s->user_work_one_lane > 0) return s->user_work_one_lane;   // translated closure with a two-valued sum: fast-forwarded
  double w = s->user_work > 0 ? s->user_work : 1e6;   // translated closure: the translator's estimate
  // row plan (lane-local re-evaluation, like the hierarchical family's): of the groups + 2 updates of a step only a few make the full pass
  if (user_rows_wanted(s, G) && user_rows_fit(s, 256, (size_t)160 * 1024)) w *= (2.0 + 0.35 * s->user_rows_groups) / (2.0 + s->user_rows_groups);
  return w;
}

// the hierarchical family's row layout (amwg_models.h: lane-local re-evaluation): a chain on one wavefront, labels that repeat with the lane
// stride, not switched off -- and the tile, the label bytes and the per-wavefront term rows must fit beside the stepper state.  The sweep kernel that
// goes with it is compiled for workgroups of at most 512 threads: a caller who ASKS for more (options.block_threads = 1024) gets the kernel that
// evaluates everything, as before round 4, instead of a "no launch geometry fits" that names the wrong cause
bool hier_rows_wanted(const amwg_sampler *s, int G) {
  return !(s->opt.block_threads > 512) && !s->user && s->model == AMWG_MODEL_HIER_NORMAL && !s->mc.group_local && s->opt.full_evaluation != 1 && G == 64 && ((s->hier_periodic_mask >> 6) & 1u) && s->d.G <= 64 && s->d.n_obs >= 64;
}
// a translated closure with a row plan (amwg_rows.h; translate.js): the same layout, LDS bytes by the same formula (UserRows<M> has HierNormalModel's)
bool user_rows_wanted(const amwg_sampler *s, int G) {
  return s->user && s->user_rows_n >= 64 && s->user_rows_groups >= 1 && s->user_rows_groups <= 64 && s->opt.full_evaluation != 1 && !(s->opt.block_threads > 512) && G == 64;
}
size_t user_rows_bytes(const amwg_sampler *s, int bt) { return HierNormalModel::rows_lds_bytes(HierNormalModel::row_pitch(s->user_rows_n), bt / 64, s->user_rows_groups); }
bool user_rows_fit(const amwg_sampler *s, int bt, size_t max_lds) {
  return bt <= 512 && lds_layout(user_rows_bytes(s, bt), s->P, bt / 64, s->pl.max_top, s->n_params).total <= max_lds;
}
// ... and the sweep prefetch with it, when the translator proved that a lane's sum depends on one entry of the swept vector (rows_sweep), the
// model has no binary parameter (BinaryStepper draws differently) and the order of every parameter vector fits the lanes of a wavefront
bool user_sweep_wanted(const amwg_sampler *s, int G, int bt, size_t max_lds) {
  return user_rows_wanted(s, G) && user_rows_fit(s, bt, max_lds) && s->user_rows_sweep && !s->user_has_binary && s->pl.max_top <= 64;
}
bool hier_rows_fit(const amwg_sampler *s, int bt, size_t max_lds) {
  const size_t data = HierNormalModel::rows_lds_bytes(HierNormalModel::row_pitch(s->d.n_obs), bt / 64, s->d.G);
  return lds_layout(data, s->P, bt / 64, s->pl.max_top, s->n_params).total <= max_lds;
}

int choose_geometry(amwg_sampler *s, int n_cus, size_t max_lds) {
  const amwg_options &o = s->opt;
  // cpb: chains per workgroup; 0 = blockDim / G.  A smaller value (one-wavefront workgroups only) is the fallback for models
  // whose per-chain state is so large that 64 / G copies do not fit LDS: the spare lane groups replicate the last chain.
  auto layout = [&](int bt, int G, int cpb = 0) {
    const size_t data_bytes = s->user ? ((user_rows_wanted(s, G) && user_rows_fit(s, bt, max_lds)) ? user_rows_bytes(s, bt) : (size_t)(G == 1 ? s->user_lds_one_lane : s->user_lds))
                              : (s->mc.group_local ? HierGlModel::gl_lds_bytes(s->d.pad, bt / 64)
                                 : ((hier_rows_wanted(s, G) && hier_rows_fit(s, bt, max_lds)) ? HierNormalModel::rows_lds_bytes(HierNormalModel::row_pitch(s->d.n_obs), bt / 64, s->d.G)
                                    : ((!s->user && s->model == AMWG_MODEL_NORMAL && G == 1) ? (normal_tile_wanted(s, bt) ? NormalModel::one_lane_tile_bytes(s->d.n_obs) : 0)
                                       : model_lds_bytes(s->model, s->d.n_obs, s->d.G, G))));
    return G > 64 ? lds_layout(data_bytes, s->P, G / 64, s->pl.max_top, s->n_params, true) : lds_layout(data_bytes, s->P, cpb ? cpb : bt / G, s->pl.max_top, s->n_params);
  };
  const int max_bt = s->user ? s->user_max_threads : model_max_threads(s->model);
  // (the hierarchical family's sweep kernel -- row layout, 64 lanes per chain -- keeps the window stream and the sweep's per-lane values in registers:
  // compiled for at most 512 threads, where a lane has 256 of them; with the 128 of a 1024-thread workgroup it ran from scratch memory, five times slower)
  // (one lane per chain with certified decisions -- the Normal family, a closure with a certified tail --: the wavefront's pass keeps 64 partial sums per lane: the 512
  // registers of a 256-thread workgroup with blocks of 16 observations, the 256 of a 512-thread one with blocks of 8 (round 6, last day: 9 spilled registers; 1.58e9 against
  // the 256-thread class's 1.55e9 at >= 131 072 chains, where 512-thread workgroups still fill every CU).  The 1024-thread class keeps the scalar-path pass (7.2e8) and is not
  // picked unless asked for; a closure's certified tail is compiled for the 256-thread class only)
  const bool cert_one_lane = (!s->user && s->model == AMWG_MODEL_NORMAL && certified_wanted(s, 1, false)) || user_cert_wanted(s, 1);
  auto fits = [&](int bt, int G) { return bt <= max_bt && bt % G == 0 && layout(bt, G).total <= max_lds && !(bt > 512 && hier_rows_wanted(s, G) && hier_rows_fit(s, 512, max_lds)) &&
                                          !(bt > 512 && user_rows_wanted(s, G) && user_rows_fit(s, 512, max_lds)) && !(bt > (user_cert_wanted(s, 1) ? 256 : 512) && G == 1 && cert_one_lane && !o.block_threads); };
  // (a closure's certified row plan -- amwg_user_sweep_cert -- ran in 256-thread workgroups for a day of round 6: in 512-thread ones it spilled 520 registers.  With the
  // S2 pass out of line -- amwg_rows.h rows_sq -- it spills 40 and the 512-thread class, two wavefronts per SIMD, is the faster one again: 2.82e9 against 1.90e9)
  if (s->user && !s->user_parallel && o.lanes_per_chain > 1)
    return fail(AMWG_EINVAL, "this closure has no loop that can be split over lanes: lanes_per_chain must be 1 (or 0 = auto), got %d", o.lanes_per_chain);
  const int bts[5] = {1024, 512, 256, 128, 64};
  int bestG = 0, bestB = 0, bestCpb = 0;
  double bestOcc = -1.0, cost1 = -1.0;
  int block1 = 0, cpb1 = 0;
  const bool fixed_lanes = o.lanes_per_chain > 0;      // (autotune_geometry calls this once per lane count, each time as a fixed one)
  for (int G = 1; G <= 1024; G <<= 1) {
    if (fixed_lanes && G != o.lanes_per_chain) continue;
    if (s->user && !s->user_parallel && G > 1) break;   // nothing to split: one lane per chain
    // translated closures: with one lane per chain every data index is wave-uniform and the compiler moves the
    // per-observation integer logic to the scalar unit, which issues 4x slower than the vector lanes (measured 2.5x on
    // the beta-Bernoulli closure); two lanes per chain keep it on the vector path at no measurable cost elsewhere
    if (s->user && s->user_parallel && !fixed_lanes && G == 1 && !(s->user_work_one_lane > 0) && !user_cert_wanted(s, 1)) continue;
    int pick = 0;
    for (int bi = 0; bi < 5; ++bi) {   // largest workgroup with >= one workgroup per CU, else the smallest that fits
      const int bt = bts[bi];
      if (o.block_threads && bt != o.block_threads) continue;
      if (G > 64 && bt != G) continue;              // a multi-wave chain is exactly one workgroup
      if (!fits(bt, G)) continue;
      pick = bt;
      if ((s->C + bt / G - 1) / (bt / G) >= n_cus) break;
    }
    int cpb = 0;
    if (!pick && G < 64 && max_bt >= 64 && (!o.block_threads || o.block_threads == 64)) {
      for (int c = 32 / G; c >= 1; c >>= 1)      // fewer chains than lane groups in a one-wavefront workgroup
        if (layout(64, G, c).total <= max_lds) { pick = 64; cpb = c; break; }
    }
    if (!pick) continue;
    const int CPB = cpb ? cpb : pick / G;
    const int64_t blocks = (s->C + CPB - 1) / CPB;
    const uint32_t lds = layout(pick, G, cpb).total;
    int64_t per_cu = 2048 / pick;                                  // 32 waves per CU
    if (lds > 0 && (int64_t)(max_lds / lds) < per_cu) per_cu = (int64_t)(max_lds / lds);
    if (per_cu < 1) per_cu = 1;
    const int64_t resident = blocks < per_cu * n_cus ? blocks : per_cu * n_cus;
    const double w_res = (double)resident * (pick / 64) / (4.0 * n_cus);          // waves a SIMD holds at once
    const double w_total = (double)blocks * (pick / 64) / (4.0 * n_cus);            // waves a SIMD has to run in all
    // stepper: 394 VALU instructions per update measured at G = 64 (rocprofv3, empty data), ~980 at G = 1, where the rnorm
    // rejection loops of the 64/G chains sharing a wave diverge (the expected maximum of 64/G geometric counts grows with
    // its logarithm); serial dependency chains, so it needs ~1.8 waves per SIMD to stay issue-bound.  Data loop: eight
    // independent terms in flight per lane, issue-bound already with one wave per SIMD.
    int lg = 0;
    for (int g = G; g < 64; g <<= 1) ++lg;
    const double S = 400.0 + 97.0 * lg;
    const double Wl = model_work(s, G) / G + (G > 64 ? 150.0 : 0.0);   // + the workgroup barrier of every evaluation
    // (round 2, hand-pipelined data loops: one wave per SIMD already issues back to back; what a lone wave loses is the issue slots
    // of its own non-arithmetic instructions, 10 % at cfg2 with one lane per chain vs four waves -- measured 3.64e8 vs 4.03e8)
    const double w1 = w_res > 1.0 ? w_res : 1.0;
    const double cost = (w_total / w_res) * (S * (w_res > 1.8 ? w_res : 1.8) + Wl * w1 * (1.0 + 0.14 / w1));
    if (G == 1) { cost1 = cost; block1 = pick; cpb1 = cpb; }
    if (bestOcc < 0 || cost < bestOcc * (1.0 - 1e-9)) { bestOcc = cost; bestG = G; bestB = pick; bestCpb = cpb; }   // bestOcc holds the best cost
  }
  // Reference order first: with ONE lane per chain a chain's log_post is summed exactly as the reference sums it (`lp += term`,
  // mcmc.js:958-960), so every draw of a seeded run is the reference's bit for bit; with more lanes only the decisions are
  // (tested), the doubles are those of the G-lane order.  Unless the caller asked for a lane count (or for AMWG_LANES_FASTEST),
  // take one lane per chain whenever the model prices it within 12 % of the cheapest geometry.
  if (o.lanes_per_chain == 0 && bestG > 1 && cost1 > 0 && cost1 <= 1.12 * bestOcc) { bestG = 1; bestB = block1; bestCpb = cpb1; }
  if (!bestG) return fail(AMWG_EINVAL, "no launch geometry fits: the model needs more than %zu bytes of LDS", max_lds);
  s->lanes = bestG;
  s->block = bestB;
  s->cpb = bestCpb;
  const int CPB = bestG > 64 ? 1 : (bestCpb ? bestCpb : bestB / bestG);
  s->grid = (int)((s->C + CPB - 1) / CPB);
  s->lds = (int)layout(bestB, bestG, bestCpb).total;
  if (s->user) {    // the kernel is compiled for this geometry afterwards
    const bool rows = user_rows_wanted(s, s->lanes) && user_rows_fit(s, s->block, max_lds);
    s->d.pad = rows ? HierNormalModel::row_pitch(s->user_rows_n) : 0;
    s->user_sweep = rows && user_sweep_wanted(s, s->lanes, s->block, max_lds);
    s->certified = s->user_sweep ? user_rows_cert_wanted(s) : user_cert_wanted(s, s->lanes);      // (amwg_user_sweep_cert / amwg_user_step_cert)
    return AMWG_OK;
  }
  const bool rows = !s->mc.group_local && hier_rows_wanted(s, s->lanes) && hier_rows_fit(s, s->block, max_lds);
  if (s->model == AMWG_MODEL_NORMAL) s->d.pad = (s->lanes == 1 && normal_tile_wanted(s, s->block)) ? 1 : 0;
  s->kernel = certified_wanted(s, s->lanes, rows) ? pick_certified_kernel(s->model, s->lanes, s->block) : nullptr;
  s->certified = s->kernel != nullptr;
  if (!s->kernel) s->kernel = s->mc.group_local ? amwg_kernel_hier_gl(s->block) : (rows ? amwg_kernel_hier_sweep(s->block) : pick_kernel(s->model, s->lanes, s->block));
  if (!s->kernel) return fail(AMWG_EINVAL, "no kernel for model %d with %d lanes per chain in workgroups of %d", s->model, s->lanes, s->block);
  return AMWG_OK;
}

// Optional tracing: roctx ranges around every burn/sample call, visible to `rocprofv3 --marker-trace`.
// the roctx library is looked up at run time; without it (or with AMWG_ROCTX=0) these are no-ops.
struct Roctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char *env = getenv("AMWG_ROCTX");
    if (env && env[0] == '0') return;
    void *h = dlopen("librocprofiler-sdk-roctx.so", RTLD_LAZY | RTLD_GLOBAL);   // ROCm 7
    if (!h) h = dlopen("libroctx64.so", RTLD_LAZY | RTLD_GLOBAL);               // older ROCm
    if (h) {
      push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
      pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
      if (!push || !pop) push = nullptr, pop = nullptr;
    }
  }
};
Roctx &roctx() { static Roctx r; return r; }

// which of the three kernels of a translated closure's code object this sampler launches (user_program)
static const char *user_kernel_symbol(const amwg_sampler *s) {
  return s->user_sweep ? (s->certified ? "amwg_user_sweep_cert" : "amwg_user_sweep") : (s->certified ? "amwg_user_step_cert" : "amwg_user_step");
}
// does this sampler's kernel decide from a model's cheaper value of log_post (amwg_kernel.h kCert: NormalModel at one lane per chain, PoisGlmModel at 16)?
static bool certified_kernel(const amwg_sampler *s) { return s->certified; }

int launch_steps(amwg_sampler *s, int64_t n, int64_t thin, double *d_draws, bool finalize = false) {
  Roctx &rx = roctx();
  if (rx.push) rx.push(d_draws ? "amwg_sample" : "amwg_burn");
  struct PopOnExit { Roctx &r; ~PopOnExit() { if (r.pop) r.pop(); } } pop_on_exit{rx};
  // a launch counts its accepted / evaluated proposals in 16-bit fields (amwg_kernel.h, TOTme): at most 65535 steps per launch
  int64_t chunk = (s->opt.steps_per_launch > 0 && s->opt.steps_per_launch < 65535) ? s->opt.steps_per_launch : 65535;
  if (d_draws && d_draws == s->d_draws && s->opt.steps_per_launch <= 0) {
    // draws that will be fetched (amwg_sample / amwg_sample_async): launches of ~32 MB of recorded rows each, so that the rows of one launch
    // leave the device while the next launches run (amwg_fetch_draws_slices); results do not depend on how a call is cut into launches
    const double per_step = (double)(s->P + s->D) * (double)s->C * 8.0 / (double)thin;
    const double steps = 33554432.0 / (per_step > 0 ? per_step : 1.0);
    if (steps < (double)chunk) chunk = steps < 16.0 ? 16 : (int64_t)steps;
  }
  // the invariants the kernel relies on, enforced where the launch is made (the kernel's own guards -- device_error -- are the backstop)
  if (s->cpb > 0 && s->block != 64) return fail(AMWG_EINVAL, "internal: %d chains per workgroup of %d threads (replicated chains need one-wavefront workgroups)", s->cpb, s->block);
  if (chunk > 65535) return fail(AMWG_EINVAL, "internal: launches of %lld steps (at most 65535)", (long long)chunk);
  StepArgs a{};
  a.C = s->C;
  a.seed = s->opt.seed;
  a.chain_offset = s->opt.chain_offset;
  a.thin = (int32_t)thin;
  a.draws = d_draws;
  a.cc = s->d_cc;
  a.is_adapting = s->d_adapt;
  a.pl = s->pl;
  a.cpb = s->cpb;
  a.sweep_update_by_update = s->opt.full_evaluation == 2 ? 1 : 0;
  a.certified = s->certified ? 1 : 0;      // (informational: certified decisions are the kernel's, not a switch inside it)
  a.bound_scale = std::ldexp(1.0, s->opt.test_bound_shift);
  a.audit_adversarial = 0;
#if defined(AMWG_AUDIT)
  { const char *e = getenv("AMWG_AUDIT_ADVERSARIAL"); a.audit_adversarial = (e && e[0] == '1') ? 1 : 0; }
#endif
  a.mc = s->mc;
  a.d = s->d;
  a.ch = s->ch;
  // (a 0-step finalize launch on chains that have stepped -- amwg_chain_diag asking for the expression's value after a certified kernel ran -- is not "the latest call":
  // the sample call's launch count, its per-launch marks and its event pair stay, so that a diag() between sample_async and fetch_draws neither loses the copy overlap
  // nor replaces the call's kernel time with its own; round-5 advisor finding)
  const bool quiet = finalize && n == 0 && s->lp_ready;
  if (!quiet) {
    s->n_launches = 0;
    s->chunk_rows.clear();
    HIP_TRY(hipEventRecord(s->ev0, s->stream));
  }
  int64_t done = 0, row = 0;
  do {
    const int64_t m = (n - done < chunk) ? n - done : chunk;
    a.n_steps = (int32_t)m;
    a.init_lp = s->lp_ready ? 0 : 1;
    a.finalize_lp = finalize ? 1 : 0;
    // steps until the first recorded step of this launch: smallest t >= 0 with (done + t) % thin == 0
    a.step0 = (thin - (done % thin)) % thin;
    a.row0 = row;
    if (s->user) {
      size_t arg_bytes = sizeof a;
      void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &arg_bytes, HIP_LAUNCH_PARAM_END};
      HIP_TRY(hipModuleLaunchKernel(s->user_fn, (unsigned)s->grid, 1, 1, (unsigned)s->block, 1, 1, (unsigned)s->lds, s->stream, nullptr, extra));
    } else {
      hipLaunchKernelGGL(s->kernel, dim3(s->grid), dim3(s->block), (size_t)s->lds, s->stream, a);
      HIP_TRY(hipGetLastError());
    }
    s->lp_ready = true;
    if (finalize || a.init_lp) s->lp_is_expression = true;                        // (the launch began / ends with the expression)
    if (m > 0 && !finalize && certified_kernel(s)) s->lp_is_expression = false;      // (it may have left the stepper's cheaper value and its bound behind)
    if (!quiet) s->n_launches++;
    if (d_draws) row += (m > a.step0) ? (m - a.step0 + thin - 1) / thin : 0;
    // a mark for amwg_fetch_draws*: only for the library's own buffer, and only once >= 8 MB of new rows (or the end of the call) stand
    // behind it -- a caller who asks for one-step launches gets a handful of events, not one per launch
    const int64_t marked = s->chunk_rows.empty() ? 0 : s->chunk_rows.back();
    if (d_draws && d_draws == s->d_draws && row > marked &&
        (done + m >= n || (double)(row - marked) * (double)(s->P + s->D) * (double)s->C * 8.0 >= 8388608.0)) {
      const size_t j = s->chunk_rows.size();
      if (j >= s->chunk_ev.size()) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        s->chunk_ev.push_back(e);
      }
      HIP_TRY(hipEventRecord(s->chunk_ev[j], s->stream));
      s->chunk_rows.push_back(row);
    }
    done += m;
  } while (done < n);
  if (!quiet) HIP_TRY(hipEventRecord(s->ev1, s->stream));
  return AMWG_OK;
}

int finish_timing(amwg_sampler *s) {
  HIP_TRY(hipEventSynchronize(s->ev1));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
  s->kernel_ms = ms;
  // what the step kernels had to say (amwg_kernel.h device_error): a launch that refused itself, or a register mirror that no longer
  // equals the state it mirrors, is an error of this call -- not a successful no-op
  if (s->ch.error) {
    int32_t bits = 0;
    HIP_TRY(hipMemcpy(&bits, s->ch.error, sizeof bits, hipMemcpyDeviceToHost));
    if (bits) {
      HIP_TRY(hipMemset(s->ch.error, 0, sizeof bits));
      return fail(AMWG_EHIP, "the step kernel reported an internal error (bits %d:%s%s%s%s): the chains' state is not to be trusted", bits,
                  (bits & 1) ? " replicated chains in a workgroup of more than one wavefront" : "", (bits & 2) ? " more than 65535 steps in one launch" : "",
                  (bits & 4) ? " the register mirror of the state is out of sync with the state" : "",
                  (bits & 8) ? " the sweep kernel was launched for a parameter vector of more than 64 entries" : "");
    }
  }
  return AMWG_OK;
}

__global__ void moments_kernel(const double *draws, int64_t rows, int P, int64_t C, double *mean, double *sd) {
  __shared__ double red[1024];
  const int p = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int64_t n = rows * C;
  double sum = 0;
  for (int64_t i = tid; i < n; i += nt) sum += draws[((i / C) * P + p) * C + (i % C)];
  red[tid] = sum;
  __syncthreads();
  for (int o = nt / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  const double m = red[0] / (double)n;
  __syncthreads();
  double ss = 0;
  for (int64_t i = tid; i < n; i += nt) { const double dlt = draws[((i / C) * P + p) * C + (i % C)] - m; ss += dlt * dlt; }
  red[tid] = ss;
  __syncthreads();
  for (int o = nt / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  if (tid == 0) { mean[p] = m; sd[p] = n > 1 ? sqrt(red[0] / (double)(n - 1)) : 0.0; }
}

// per chain and recorded value: mean and (n-1) variance of each half of the chain's kept draws
// out[((h*2 + stat) * PR + p) * C + c], stat 0 = mean, 1 = variance; draws [row][PR][C] (coalesced over chains)
__global__ void chain_halves_kernel(const double *draws, int64_t rows, int PR, int64_t C, double *out) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int p = blockIdx.y;
  if (c >= C) return;
  const int64_t half = rows / 2;
  for (int h = 0; h < 2; ++h) {
    const int64_t r0 = h * half, r1 = r0 + half;
    double m = 0, m2 = 0;   // Welford
    for (int64_t r = r0; r < r1; ++r) {
      const double x = draws[(r * PR + p) * C + c];
      const double dlt = x - m;
      m += dlt / (double)(r - r0 + 1);
      m2 += dlt * (x - m);
    }
    out[((size_t)(h * 2 + 0) * PR + p) * C + c] = m;
    out[((size_t)(h * 2 + 1) * PR + p) * C + c] = half > 1 ? m2 / (double)(half - 1) : 0.0;
  }
}

// 8 independent fma chains per lane, no memory traffic: the fp64 issue rate the chip sustains
__global__ void __launch_bounds__(1024) fp64_peak_kernel(double *out, int iters, double a, double b) {
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
      x4 = __builtin_fma(x4, a, b); x5 = __builtin_fma(x5, a, b); x6 = __builtin_fma(x6, a, b); x7 = __builtin_fma(x7, a, b);
    }
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
}

}  // namespace

int amwg_fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

extern "C" {

const char *amwg_last_error(void) { return g_err.c_str(); }
// which sources this binary was built from (tools/build_id.py: a hash over every source of the library, and one over the device sources + compiler flags)
const char *amwg_version(void) { return "amwg-mi355x 0.4 (gfx950) build " AMWG_BUILD_ID " kernels " AMWG_KERNEL_ID; }

double amwg_exp(double x) { return exp_v8(x); }
double amwg_log(double x) { return log_v8(x); }
#if defined(AMWG_SELFTEST)      // include/amwg_selftest.h: the building blocks one by one, for the test suite (libamwg_selftest.so)
double amwg_pow(double x, double y) { return pow_v8(x, y); }
double amwg_log1p(double x) { return log1p_v8(x); }
double amwg_expm1(double x) { return expm1_v8(x); }
double amwg_math1(int32_t fn, double x) { return math1_by_id(fn, x); }
double amwg_math2(int32_t fn, double x, double y) {
  switch (fn) {
    case 0: return atan2_v8(x, y);
    case 1: return hypot2_v8(x, y);
    case 2: return js_mod(x, y);                 // JavaScript's `%`
    case 3: return (double)js_toint32(x);        // `x | 0`
  }
  return __builtin_nan("");
}
double amwg_hypot3(double x, double y, double z) { return hypot3_v8(x, y, z); }
double amwg_ld_host(int32_t id, double x, double a, double b, double c) { return ld_by_id(id, x, a, b, c); }
#endif
double amwg_uniform(uint64_t seed, uint64_t chain, uint64_t index) {
  ChainStream s;
  s.init(seed, chain, index);
  return s.next();
}

}  // extern "C"

// The data-only tables of two_valued_sum (amwg_models.h), back to back: the observations as bits (w), the ones before every
// word (pre), and for each symbol the marks of the occurrences whose immediately preceding run of the OTHER symbol has odd
// length (om1 / om0) with their per-word prefix counts (po1 / po0).
static std::vector<uint32_t> two_valued_tables(const uint8_t *xb, int N) {
  const size_t W = two_valued_words(N);
  std::vector<uint32_t> tab(6 * W, 0u);
  for (int i = 0; i < N; ++i) if (xb[i]) tab[(size_t)i >> 5] |= 1u << (i & 31);
  for (size_t k = 1; k < W; ++k) tab[W + k] = tab[W + k - 1] + (uint32_t)__builtin_popcount(tab[k - 1]);
  for (int sym = 1; sym >= 0; --sym) {
    uint32_t *om = tab.data() + (sym ? 2 : 4) * W, *po = om + W;
    int run = 0;   // length of the current run of the other symbol
    for (int i = 0; i < N; ++i) {
      const int v = xb[i] ? 1 : 0;
      if (v == sym) { if (run & 1) om[(size_t)i >> 5] |= 1u << (i & 31); run = 0; } else ++run;
    }
    for (size_t k = 1; k < W; ++k) po[k] = po[k - 1] + (uint32_t)__builtin_popcount(om[k - 1]);
  }
  return tab;
}

#if defined(AMWG_SELFTEST)
// tests only: thread j sums the same bit sequence from acc0[j] with addends l1[j], l0[j], once with two_valued_sum and
// once term by term
__global__ void two_valued_check_kernel(const uint32_t *tab, int N, int64_t m, const double *acc0, const double *l1, const double *l0,
                                        double *out_ff, double *out_seq) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const size_t W = BetaBernModel::words(N);
  BitData B{tab, tab + W, tab + 2 * W, tab + 3 * W, tab + 4 * W, tab + 5 * W, N};
  out_ff[j] = two_valued_sum(acc0[j], l1[j], l0[j], B);
  double acc = acc0[j];
  for (int i = 0; i < N; ++i) acc = acc + (((tab[i >> 5] >> (i & 31)) & 1u) ? l1[j] : l0[j]);
  out_seq[j] = acc;
}
#endif

// ---- pieces of construction shared by the built-in and the translated models
#define TRYB(x) do { int rc_ = (x); if (rc_ != AMWG_OK) return rc_; } while (0)
#define HIPB(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(AMWG_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

static int check_options(const amwg_options *options, int max_threads) {
  if (options->chains < 1) return fail(AMWG_EINVAL, "amwg_create: chains must be >= 1");
  // a launch counts its accepted / evaluated proposals in 16-bit fields: longer launches are not silently shortened (launch_info and the
  // bench's per-launch figures assume the requested size)
  if (options->steps_per_launch < 0 || options->steps_per_launch > 65535)
    return fail(AMWG_EINVAL, "steps_per_launch must be 0 (auto) or 1..65535, got %d", options->steps_per_launch);
  const int G_opt = options->lanes_per_chain;
  if (G_opt && G_opt != AMWG_LANES_FASTEST && G_opt != AMWG_LANES_AUTOTUNE && (G_opt < 1 || G_opt > 1024 || (G_opt & (G_opt - 1))))
    return fail(AMWG_EINVAL, "lanes_per_chain must be a power of two in 1..1024 (or 0 = auto, AMWG_LANES_FASTEST = -1, AMWG_LANES_AUTOTUNE = -2)");
  if (G_opt > 64 && options->block_threads && options->block_threads != G_opt)
    return fail(AMWG_EINVAL, "a chain on %d lanes is one workgroup of %d threads: block_threads must be 0 or %d", G_opt, G_opt, G_opt);
  if (G_opt > max_threads) return fail(AMWG_EINVAL, "lanes_per_chain %d exceeds this model's workgroup limit %d", G_opt, max_threads);
  if (options->block_threads && (options->block_threads % 64 || options->block_threads > 1024 || options->block_threads < 64))
    return fail(AMWG_EINVAL, "block_threads must be a multiple of 64 in 64..1024");
  if (options->block_threads > max_threads)
    return fail(AMWG_EINVAL, "block_threads %d exceeds this model's workgroup limit %d", options->block_threads, max_threads);
#if defined(AMWG_AUDIT)      // (the audit build also SHRINKS the bounds -- the converse experiment of tools/bound_audit.py: how far before a chain differs)
  if (options->test_bound_shift < -60 || options->test_bound_shift > 40) return fail(AMWG_EINVAL, "test_bound_shift must be -60..40 in the audit build, got %d", options->test_bound_shift);
#else
  if (options->test_bound_shift < 0 || options->test_bound_shift > 40) return fail(AMWG_EINVAL, "test_bound_shift must be 0..40, got %d", options->test_bound_shift);
#endif
  if (options->sufficient_statistics != 0 && options->sufficient_statistics != 1) return fail(AMWG_EINVAL, "sufficient_statistics must be 0 or 1, got %d", options->sufficient_statistics);
  if (options->sufficient_statistics && (options->full_evaluation != 0 || options->exact_division)) return fail(AMWG_EINVAL, "sufficient_statistics decides from certified values: not with full_evaluation or exact_division");
  if (options->full_evaluation < 0 || options->full_evaluation > 2)
    return fail(AMWG_EINVAL, "full_evaluation must be 0 (default), 1 (every evaluation passes over all the data) or 2 (sweeps decided update by update), got %d", options->full_evaluation);
  return AMWG_OK;
}

// completed params (mcmc.js:357-403) -> flat layout.  Stepped parameters first (s->n_params of them, any number up to kMaxIndex);
// trailing AMWG_FIXED entries only add state slots.  The per-parameter table goes to the device (upload_layout).
static int build_layout(amwg_sampler *s, const amwg_param_desc *params, int n_params, bool allow_fixed) {
  ParamLayout &pl = s->pl;
  pl.max_top = 1;
  int P = 0, n_stepped = 0;
  std::vector<int32_t> base, len, top, multidim;
  bool fixed_seen = false;
  for (int p = 0; p < n_params; ++p) {
    const amwg_param_desc &q = params[p];
    if (q.type == AMWG_FIXED) {
      if (!allow_fixed) return fail(AMWG_EINVAL, "parameter %d: AMWG_FIXED entries are only supported by amwg_create_user", p);
      if (q.len < 1) return fail(AMWG_EINVAL, "parameter %d: bad len %d", p, q.len);
      fixed_seen = true;
      P += q.len;
      continue;
    }
    if (fixed_seen) return fail(AMWG_EINVAL, "parameter %d: stepped parameters must come before the AMWG_FIXED entries", p);
    if (q.type != AMWG_REAL && q.type != AMWG_INT && q.type != AMWG_BINARY)
      return fail(AMWG_EINVAL, "AmwgStepper can't handle parameter %d with type %d", p, q.type);   // mcmc.js:867
    // the built-in families' kernels are compiled without the BinaryStepper branch (none of them has a binary parameter): a binary
    // parameter there would silently be stepped by the Metropolis stepper instead of mcmc.js:753-767 -- refuse it
    if (q.type == AMWG_BINARY && !allow_fixed)
      return fail(AMWG_EINVAL, "parameter %d: the built-in model families have no binary parameters (BinaryStepper runs for translated closures, amwg_create_user)", p);
    if (n_stepped >= kMaxIndex) return fail(AMWG_EINVAL, "more than %d stepped parameters", kMaxIndex);
    if (q.len < 1 || q.top < 1 || q.len % q.top) return fail(AMWG_EINVAL, "parameter %d: bad dim (len %d, top %d)", p, q.len, q.top);
    if (q.top > kMaxIndex) return fail(AMWG_EINVAL, "parameter %d: leading dimension %d > %d", p, q.top, kMaxIndex);
    if (!q.multidim && q.len != 1) return fail(AMWG_EINVAL, "parameter %d: dim [1] but len %d", p, q.len);
    base.push_back(P); len.push_back(q.len); top.push_back(q.top); multidim.push_back(q.multidim ? 1 : 0);
    if (q.multidim && q.top > pl.max_top) pl.max_top = q.top;
    P += q.len;
    ++n_stepped;
    pl.P_stepped = P;
  }
  if (n_stepped < 1) return fail(AMWG_EINVAL, "no parameter to step");
  pl.n_params = n_stepped;
  pl.P = P;
  s->P = P;
  s->n_params = n_stepped;
  s->h_layout.clear();
  for (const std::vector<int32_t> *v : {&base, &len, &top, &multidim}) s->h_layout.insert(s->h_layout.end(), v->begin(), v->end());
  return AMWG_OK;
}

// AMWG_TIMING=1: the constructor's phases on stderr (development aid; the end-to-end bench reports the constructor as a whole)
namespace {
struct PhaseClock {
  bool on;
  std::chrono::steady_clock::time_point t0;
  PhaseClock() : on(getenv("AMWG_TIMING") && getenv("AMWG_TIMING")[0] == '1'), t0(std::chrono::steady_clock::now()) {}
  void mark(const char *what) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[amwg timing] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};
}  // namespace
static int open_device(amwg_sampler *s, hipDeviceProp_t *prop) {
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev < 1) return fail(AMWG_EHIP, "no HIP device available (%s)", hipGetErrorString(e));
  if (s->device < 0 || s->device >= ndev) return fail(AMWG_EINVAL, "device %d out of range (%d visible)", s->device, ndev);
  HIPB(hipSetDevice(s->device));
  HIPB(hipGetDeviceProperties(prop, s->device));
  HIPB(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  HIPB(hipEventCreate(&s->ev0));
  HIPB(hipEventCreate(&s->ev1));
  return AMWG_OK;
}

// per-component constants and per-chain state (every chain starts at the same init, mcmc.js:954-957)
static int alloc_chain_state(amwg_sampler *s, const amwg_param_desc *params, int n_params, const double *init,
                             const amwg_comp_opt *comp_opts) {
  const int P = s->P;
  std::vector<CompConst> hcc((size_t)P);
  s->h_adapt.resize((size_t)P);
  for (int p = 0, ci = 0; p < n_params; ++p)
    for (int e2 = 0; e2 < params[p].len; ++e2, ++ci) {
      const amwg_comp_opt &o = comp_opts[ci];
      hcc[ci] = CompConst{params[p].lower, params[p].upper, o.max_adaptation, o.initial_adaptation, o.target_accept_rate,
                          o.batch_size, params[p].type};
      s->h_adapt[ci] = o.is_adapting ? 1 : 0;
    }
  TRYB(dev_alloc(s, &s->d_cc, (size_t)P));
  HIPB(hipMemcpy(s->d_cc, hcc.data(), (size_t)P * sizeof(CompConst), hipMemcpyHostToDevice));
  TRYB(dev_alloc(s, &s->d_adapt, (size_t)P));
  HIPB(hipMemcpy(s->d_adapt, s->h_adapt.data(), (size_t)P, hipMemcpyHostToDevice));
  {
    int32_t *d_tab = nullptr;
    TRYB(dev_alloc(s, &d_tab, s->h_layout.size()));
    HIPB(hipMemcpy(d_tab, s->h_layout.data(), s->h_layout.size() * 4, hipMemcpyHostToDevice));
    s->pl.tab = d_tab;
  }

  const size_t PC = (size_t)P * (size_t)s->C, C = (size_t)s->C;
  ChainArrays &ch = s->ch;
  TRYB(dev_alloc(s, &ch.state, PC));
  TRYB(dev_alloc(s, &ch.prop_log_scale, PC));
  TRYB(dev_alloc(s, &ch.acceptance_count, PC));
  TRYB(dev_alloc(s, &ch.iterations_since_adaption, PC));
  TRYB(dev_alloc(s, &ch.batch_count, PC));
  TRYB(dev_alloc(s, &ch.accepts, PC));
  TRYB(dev_alloc(s, &ch.inbounds, PC));
  const bool wide_perm = s->n_params > kPackedNamed;
  TRYB(dev_alloc(s, &ch.perm, C));
  ch.perm16 = nullptr;
  if (wide_perm) TRYB(dev_alloc(s, &ch.perm16, (size_t)s->n_params * C));
  TRYB(dev_alloc(s, &ch.rng_n, C));
  TRYB(dev_alloc(s, &ch.lp_curr, C));
  TRYB(dev_alloc(s, &ch.lp_eps, C));
  // (64 doubles per wavefront of a one-lane-per-chain launch: where the certified pass of the Normal family / a closure's certified tail leaves the wavefront's means for the
  // scalar memory path, amwg_pass.h norm_sq_pass_wave.  C / 64 wavefronts + the last workgroup's spare ones; AMWG_WAVE_SCRATCH=0: none, the pass broadcasts with v_readlane)
  s->d.wave_scratch = nullptr;
  {
    const char *env = getenv("AMWG_WAVE_SCRATCH");
    if (!(env && env[0] == '0')) TRYB(dev_alloc(s, &s->d.wave_scratch, ((size_t)C / 64 + 64) * 64));
  }
  TRYB(dev_alloc(s, &ch.error, (size_t)1));
  ch.audit = nullptr;
  ch.audit_hist = nullptr;
#if defined(AMWG_AUDIT) || defined(AMWG_X_PHASES)      // (libamwg_audit.so: the bound audit's per-chain maxima and histograms, amwg_kernel.h "BOUND AUDIT"; the phase-clock development build)
  TRYB(dev_alloc(s, &ch.audit, 4 * C));
  TRYB(dev_alloc(s, &ch.audit_hist, (size_t)128));
  HIPB(hipMemset(ch.audit, 0, 4 * C * 8));
  HIPB(hipMemset(ch.audit_hist, 0, 128 * 8));
#endif
  {
    std::vector<double> tmp(PC);
    for (int p = 0; p < P; ++p) for (size_t c = 0; c < C; ++c) tmp[(size_t)p * C + c] = init[p];
    HIPB(hipMemcpy(ch.state, tmp.data(), PC * 8, hipMemcpyHostToDevice));
    for (int p = 0; p < P; ++p) for (size_t c = 0; c < C; ++c) tmp[(size_t)p * C + c] = comp_opts[p].prop_log_scale;
    HIPB(hipMemcpy(ch.prop_log_scale, tmp.data(), PC * 8, hipMemcpyHostToDevice));
    uint64_t ident = 0;
    for (int i = 0; i < kPackedNamed; ++i) ident |= (uint64_t)i << (4 * i);
    std::vector<uint64_t> pv(C, ident);
    HIPB(hipMemcpy(ch.perm, pv.data(), C * 8, hipMemcpyHostToDevice));
    if (wide_perm) {   // initial order = Object.keys(params) (mcmc.js:839)
      std::vector<uint16_t> p16((size_t)s->n_params * C);
      for (int k = 0; k < s->n_params; ++k) for (size_t c = 0; c < C; ++c) p16[(size_t)k * C + c] = (uint16_t)k;
      HIPB(hipMemcpy(ch.perm16, p16.data(), p16.size() * 2, hipMemcpyHostToDevice));
    }
  }
  HIPB(hipMemset(ch.acceptance_count, 0, PC * 4));
  HIPB(hipMemset(ch.iterations_since_adaption, 0, PC * 4));
  HIPB(hipMemset(ch.batch_count, 0, PC * 4));
  HIPB(hipMemset(ch.accepts, 0, PC * 4));
  HIPB(hipMemset(ch.inbounds, 0, PC * 4));
  HIPB(hipMemset(ch.rng_n, 0, C * 8));
  HIPB(hipMemset(ch.lp_curr, 0, C * 8));
  HIPB(hipMemset(ch.lp_eps, 0, C * 8));
  HIPB(hipMemset(ch.error, 0, sizeof(int32_t)));
  return AMWG_OK;
}

// ---- hiprtc: a translated closure + the step kernel -> code object for one (lanes, workgroup) geometry
static std::string user_program(const char *source, int lanes, int block) {
  std::string p = "#include \"amwg_kernel.h\"\n#include \"amwg_user.h\"\n";
  p += source;
  char tail[1200];
  // amwg_user_step: the step kernel for this geometry.  amwg_user_sweep: for a closure with a row plan (amwg_rows.h: UserModel::kLaneReuse) on a whole
  // wavefront per chain, the same stepper with the sweep prefetch (amwg_sweep_kernel's twin); an empty kernel otherwise -- the host never launches it then.
  snprintf(tail, sizeof tail,
           "\nextern \"C\" __global__ void __launch_bounds__(%d) amwg_user_step(const amwg::StepArgs a) {\n"
           "  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];\n"
           "  amwg::step_body<amwg::UserModel, %d>(a, smem);\n}\n"
           "extern \"C\" __global__ void __launch_bounds__(%d) amwg_user_sweep(const amwg::StepArgs a) {\n"
           "  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];\n"
           "  if constexpr (amwg::LaneReuseOf<amwg::UserModel>::value && %d == 64 && %d <= 512) amwg::step_body<amwg::UserModel, 64, 512, false, true>(a, smem);\n}\n",
           block, lanes, block, lanes, block);
  p += tail;
  // amwg_user_step_cert: for a closure with a certified tail (amwg_user.h norm_tail_approx: UserModel::kCertified) at the lane count it has one for, the stepper
  // that decides accept tests from it (amwg_step_kernel_cert's twin; BT: the wavefront's pass needs the 512 registers of a workgroup of at most 256 threads)
  // amwg_user_sweep_cert: a row plan the translator marked kRowCert (amwg_rows.h: certified values + the expression in the reference's order): amwg_sweep_kernel_cert's twin
  snprintf(tail, sizeof tail,
           "extern \"C\" __global__ void __launch_bounds__(%d) amwg_user_sweep_cert(const amwg::StepArgs a) {\n"
           "  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];\n"
           "  if constexpr (amwg::LaneReuseOf<amwg::UserModel>::value && amwg::CertifiedAt<amwg::UserModel, 64>::value && amwg::CertNeedsRows<amwg::UserModel>::value && %d == 64 && %d <= 512)\n"
           "    amwg::step_body<amwg::UserModel, 64, 512, false, true, true>(a, smem);\n}\n",
           block, lanes, block);
  p += tail;
  snprintf(tail, sizeof tail,
           "extern \"C\" __global__ void __launch_bounds__(%d) amwg_user_step_cert(const amwg::StepArgs a) {\n"
           "  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];\n"
           "  if constexpr (amwg::CertifiedAt<amwg::UserModel, %d>::value && !amwg::CertNeedsRows<amwg::UserModel>::value) amwg::step_body<amwg::UserModel, %d, %d, false, false, true>(a, smem);\n}\n",
           block, lanes, lanes, block <= 256 ? 256 : 1024);
  p += tail;
  return p;
}

// ---- on-disk cache of compiled code objects.  hiprtc takes ~0.6 s per closure and geometry; the reference's own use -- one chain, a
// script run once (README.md:41-42) -- would pay that at every start.  Key = everything that determines the code object: the program
// text, the embedded kernel headers, the compile options, the target, the hiprtc version.  Files: <dir>/<128-bit key hash>.hsaco, each
// carrying the key's length and a second hash, written to a temporary name and renamed (concurrent processes never see a partial file).
// Directory: $AMWG_CACHE_DIR, else $XDG_CACHE_HOME/amwg, else $HOME/.cache/amwg; AMWG_CACHE_DIR="" (empty) or an unwritable directory
// disables it silently -- the cache is an optimisation, never a requirement.
namespace {
struct CacheKey { uint64_t h1, h2, h3; uint64_t len; };
CacheKey hash_key(const std::vector<std::string> &parts) {
  CacheKey k{0xcbf29ce484222325ull, 0x84222325cbf29ce4ull, 0x9e3779b97f4a7c15ull, 0};
  for (const std::string &p : parts) {
    for (unsigned char c : p) {
      k.h1 = (k.h1 ^ c) * 0x100000001b3ull;                                   // FNV-1a
      k.h2 = (k.h2 + c + (k.h2 << 6) + (k.h2 >> 2)) * 0xff51afd7ed558ccdull;  // an unrelated mix
      k.h3 = ((k.h3 << 5) | (k.h3 >> 59)) ^ (c * 0xc4ceb9fe1a85ec53ull);
    }
    k.h1 = (k.h1 ^ 0xff) * 0x100000001b3ull;                                  // part separator
    k.len += p.size() + 1;
  }
  return k;
}
std::string cache_dir() {
  if (const char *d = getenv("AMWG_CACHE_DIR")) return d;       // (empty string: disabled)
  if (const char *x = getenv("XDG_CACHE_HOME")) if (*x) return std::string(x) + "/amwg";
  if (const char *h = getenv("HOME")) if (*h) return std::string(h) + "/.cache/amwg";
  return "";
}
void make_dirs(const std::string &path) {      // mkdir -p; the cache directory itself is private to the user (code objects are loaded from it)
  for (size_t i = 1; i <= path.size(); ++i)
    if (i == path.size() || path[i] == '/') (void)mkdir(path.substr(0, i).c_str(), i == path.size() ? 0700 : 0755);
}
// Code objects are LOADED from this directory: it is used only while it belongs to this user and nobody else can write to it.  An existing
// directory with group / other write bits is tightened to 0700 when it is ours (a leftover 0755 from before round 4 included); one that
// belongs to somebody else, or is not a directory (a symlink is not followed), switches the cache off.  The payload sum in a file's header
// guards against damage, not against a planted file -- this check, O_NOFOLLOW and the owner test in cache_read are what guard against that.
bool cache_dir_trusted(const std::string &dir) {
  struct stat st;
  if (lstat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != geteuid()) return false;
  if ((st.st_mode & (S_IWGRP | S_IWOTH)) != 0 && chmod(dir.c_str(), 0700) != 0) return false;
  return true;
}
uint64_t payload_sum(const std::vector<char> &code) {      // FNV-1a over the code bytes: a damaged file is recompiled, not handed to the loader
  uint64_t h = 0xcbf29ce484222325ull;
  for (unsigned char c : code) h = (h ^ c) * 0x100000001b3ull;
  return h;
}
const char kCacheMagic[8] = {'A', 'M', 'W', 'G', 'c', 'o', '0', '2'};
bool cache_read(const std::string &file, const CacheKey &k, std::vector<char> *code) {
  const int fd = open(file.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
  if (fd < 0) return false;
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH)) != 0) { close(fd); return false; }      // not ours: not loaded (and not removed)
  FILE *f = fdopen(fd, "rb");
  if (!f) { close(fd); return false; }
  char magic[8];
  uint64_t hdr[4] = {0, 0, 0, 0}, n = 0;
  bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, kCacheMagic, 8) == 0 && fread(hdr, 8, 4, f) == 4 && fread(&n, 8, 1, f) == 1 &&
            hdr[0] == k.h2 && hdr[1] == k.h3 && hdr[2] == k.len && n > 0 && n < (1ull << 31);
  if (ok) { code->resize((size_t)n); ok = fread(code->data(), 1, (size_t)n, f) == (size_t)n && payload_sum(*code) == hdr[3]; }
  fclose(f);
  if (!ok) (void)remove(file.c_str());      // stale format, truncated or damaged: gone, the caller compiles
  return ok;
}
void cache_write(const std::string &dir, const std::string &file, const CacheKey &k, const std::vector<char> &code) {
  make_dirs(dir);
  if (!cache_dir_trusted(dir)) return;
  const std::string tmp = file + ".tmp." + std::to_string((long)getpid());
  const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
  if (fd < 0) return;
  FILE *f = fdopen(fd, "wb");
  if (!f) { close(fd); (void)remove(tmp.c_str()); return; }
  const uint64_t hdr[4] = {k.h2, k.h3, k.len, payload_sum(code)}, n = code.size();
  const bool ok = fwrite(kCacheMagic, 1, 8, f) == 8 && fwrite(hdr, 8, 4, f) == 4 && fwrite(&n, 8, 1, f) == 1 && fwrite(code.data(), 1, code.size(), f) == code.size();
  if (fclose(f) != 0 || !ok || rename(tmp.c_str(), file.c_str()) != 0) (void)remove(tmp.c_str());
}
std::atomic<int> g_cache_hits{0}, g_cache_misses{0};      // (samplers may be created from several threads)
}  // namespace

static void dump_code_object(const std::vector<char> &code) {      // development aid: inspect the ISA with llvm-objdump
  if (const char *dump = getenv("AMWG_DUMP_CODE_OBJECT")) {
    if (FILE *f = fopen(dump, "wb")) { fwrite(code.data(), 1, code.size(), f); fclose(f); }
  }
}

// use_cache = false: compile even if the on-disk cache has the object (the caller found the cached one unloadable)
static int compile_user(const char *source, int lanes, int block, const char *arch, std::vector<char> *code, bool use_cache = true) {
  static const char *names[] = {"amwg_stdint.h", "amwg_types.h", "amwg_math.h", "amwg_div.h", "amwg_ld.h", "amwg_philox.h",
                                "amwg_kernel.h", "amwg_user.h", "amwg_twoval.h", "amwg_kval.h", "amwg_trig.h", "amwg_pass.h", "amwg_rows.h", "amwg_window.h", "amwg_ptail.h"};
  const char *texts[] = {amwg_hdr_stdint, amwg_hdr_types, amwg_hdr_math, amwg_hdr_div, amwg_hdr_ld, amwg_hdr_philox,
                         amwg_hdr_kernel, amwg_hdr_user, amwg_hdr_twoval, amwg_hdr_kval, amwg_hdr_trig, amwg_hdr_pass, amwg_hdr_rows, amwg_hdr_window, amwg_hdr_ptail};
  constexpr int kHeaders = (int)(sizeof(texts) / sizeof(texts[0]));
  const std::string prog_src = user_program(source, lanes, block);
#if defined(AMWG_AUDIT)      // (libamwg_audit.so: the certified kernels of translated closures record |A - E| / eps as the built-in families' do)
  const char *const kOpts[] = {"-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-value", "-DAMWG_AUDIT=1"};
#elif defined(AMWG_X_PHASES)      // (the phase-clock development build: translated closures are clocked too)
  const char *const kOpts[] = {"-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-value", "-DAMWG_X_PHASES=1"};
#else
  const char *const kOpts[] = {"-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-value", "-falign-loops=64"};
#endif
  // the on-disk cache (see above)
  std::string dir = cache_dir(), file;
  CacheKey key{};
  if (!dir.empty()) {
    int rt_major = 0, rt_minor = 0;
    (void)hiprtcVersion(&rt_major, &rt_minor);
    std::vector<std::string> parts = {prog_src, arch, "hiprtc " + std::to_string(rt_major) + "." + std::to_string(rt_minor)};
    for (const char *o : kOpts) parts.push_back(o);
    for (const char *t : texts) parts.push_back(t);
    key = hash_key(parts);
    char name[64];
    snprintf(name, sizeof name, "/%016llx%016llx.hsaco", (unsigned long long)key.h1, (unsigned long long)key.h2);
    file = dir + name;
    if (use_cache && cache_dir_trusted(dir) && cache_read(file, key, code)) { ++g_cache_hits; dump_code_object(*code); return AMWG_OK; }
    if (!use_cache) (void)remove(file.c_str());
  }
  ++g_cache_misses;
  hiprtcProgram prog = nullptr;
  hiprtcResult r = hiprtcCreateProgram(&prog, prog_src.c_str(), "amwg_user_model.hip", kHeaders, texts, names);
  if (r != HIPRTC_SUCCESS) return fail(AMWG_EHIP, "hiprtcCreateProgram failed: %s", hiprtcGetErrorString(r));
  const std::string arch_opt = std::string("--offload-arch=") + arch;
  // same floating-point contract as the Makefile: one rounding per operation, no fused contraction
  std::vector<const char *> opts = {arch_opt.c_str()};
  for (const char *o : kOpts) opts.push_back(o);
  r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
  if (r != HIPRTC_SUCCESS) {
    size_t n = 0;
    hiprtcGetProgramLogSize(prog, &n);
    std::string log(n ? n : 1, '\0');
    if (n) hiprtcGetProgramLog(prog, &log[0]);
    hiprtcDestroyProgram(&prog);
    g_err = "the translated log_post did not compile (hiprtc): " + log;
    return AMWG_EINVAL;
  }
  size_t cs = 0;
  hiprtcGetCodeSize(prog, &cs);
  code->resize(cs);
  hiprtcGetCode(prog, code->data());
  hiprtcDestroyProgram(&prog);
  if (!file.empty()) cache_write(dir, file, key, *code);
  dump_code_object(*code);
  return AMWG_OK;
}

// ---- AMWG_LANES_AUTOTUNE: measure instead of model.  Every lane count whose geometry fits is prepared (`prepare`: kernel lookup or
// hiprtc compile + load, plus whatever depends on the lane count), run for a few steps on the real chain state -- which is saved
// before and restored after, so tuning leaves no trace in the chains -- and timed with HIP events.  The fastest wins, except that one
// lane per chain (the reference's summation order) is kept whenever it MEASURES within 12 % of the fastest.
struct TuneCandidate { int lanes, block, grid, lds, cpb; step_kernel_t kernel; hipModule_t module; hipFunction_t fn; float ms; int pad; bool sweep; };

template <class Prepare>
static int autotune_geometry(amwg_sampler *s, int n_cus, size_t max_lds, Prepare prepare) {
  // the chain state the timing runs touch
  const size_t PC = (size_t)s->P * (size_t)s->C, C = (size_t)s->C;
  std::vector<std::pair<void *, size_t>> parts = {
      {s->ch.state, PC * 8}, {s->ch.prop_log_scale, PC * 8}, {s->ch.acceptance_count, PC * 4}, {s->ch.iterations_since_adaption, PC * 4},
      {s->ch.batch_count, PC * 4}, {s->ch.accepts, PC * 4}, {s->ch.inbounds, PC * 4}, {s->ch.perm, C * 8}, {s->ch.rng_n, C * 8}, {s->ch.lp_curr, C * 8}, {s->ch.lp_eps, C * 8}};
  if (s->ch.perm16) parts.push_back({s->ch.perm16, (size_t)s->n_params * C * 2});
  size_t total = 0;
  for (auto &p : parts) total += (p.second + 255) & ~(size_t)255;
  DevBuf save;
  HIP_TRY(save.alloc(total));
  auto copy_all = [&](bool restore) -> hipError_t {
    size_t off = 0;
    for (auto &p : parts) {
      char *sv = save.as<char>() + off;
      hipError_t e = restore ? hipMemcpyAsync(p.first, sv, p.second, hipMemcpyDeviceToDevice, s->stream) : hipMemcpyAsync(sv, p.first, p.second, hipMemcpyDeviceToDevice, s->stream);
      if (e != hipSuccess) return e;
      off += (p.second + 255) & ~(size_t)255;
    }
    return hipStreamSynchronize(s->stream);
  };
  HIP_TRY(copy_all(false));
  const int32_t wanted = s->opt.lanes_per_chain;
  std::vector<TuneCandidate> cand;
  std::string first_error;
  for (int G = 1; G <= 1024; G <<= 1) {
    s->opt.lanes_per_chain = G;
    if (choose_geometry(s, n_cus, max_lds) != AMWG_OK || prepare() != AMWG_OK) { if (first_error.empty()) first_error = g_err; continue; }
    TuneCandidate c{s->lanes, s->block, s->grid, s->lds, s->cpb, s->kernel, s->user_module, s->user_fn, 0.f, s->d.pad, s->user_sweep};
    // One untimed launch first (it evaluates log_post(init), stages the data for the first time and warms the instruction cache); then the
    // run length is scaled until a launch takes >= 1 ms -- short data loops would otherwise be ranked by launch overhead and noise -- and the
    // candidate's figure is the FASTEST of three such launches, per step.  The runs continue the chains from one another (a valid lp_curr,
    // no init evaluation inside a timed launch); the saved state is put back once the candidate is done.
    bool ok = true;
    s->lp_ready = false;
    ok = launch_steps(s, 1, 1, nullptr) == AMWG_OK && finish_timing(s) == AMWG_OK;
    int n_tune = 3, kept = 0;
    float best = -1.f;
    for (int rep = 0; rep < 10 && ok && kept < 3; ++rep) {
      ok = launch_steps(s, n_tune, 1, nullptr) == AMWG_OK && finish_timing(s) == AMWG_OK;
      if (!ok) break;
      const float ms = (float)s->kernel_ms;
      if (ms < 1.0f && n_tune < 192) { n_tune *= 4; continue; }      // too short to rank: a longer run
      const float per_step = ms / (float)n_tune;
      if (best < 0 || per_step < best) best = per_step;
      ++kept;
    }
    if (copy_all(true) != hipSuccess) ok = false;
    c.ms = best;
    ok = ok && best >= 0;
    s->lp_ready = false;
    if (ok) cand.push_back(c);
    else if (c.module) { (void)hipModuleUnload(c.module); }
    s->user_module = nullptr;
    s->user_fn = nullptr;
  }
  s->opt.lanes_per_chain = wanted;
  if (cand.empty()) return fail(AMWG_EINVAL, "autotune: no lane count could be run (%s)", first_error.c_str());
  size_t best = 0;
  for (size_t i = 1; i < cand.size(); ++i) if (cand[i].ms < cand[best].ms) best = i;
  if (cand[0].lanes == 1 && cand[0].ms <= 1.12f * cand[best].ms) best = 0;      // reference order first
  for (size_t i = 0; i < cand.size(); ++i) if (i != best && cand[i].module) (void)hipModuleUnload(cand[i].module);
  const TuneCandidate &c = cand[best];
  s->lanes = c.lanes; s->block = c.block; s->grid = c.grid; s->lds = c.lds; s->cpb = c.cpb; s->kernel = c.kernel; s->user_module = c.module; s->user_fn = c.fn;
  s->certified = s->user ? (c.sweep ? user_rows_cert_wanted(s) : user_cert_wanted(s, c.lanes)) : (c.kernel != nullptr && c.kernel == pick_certified_kernel(s->model, c.lanes, c.block));
  s->d.pad = c.pad; s->user_sweep = c.sweep;      // (the row layout and the sweep kernel go with the geometry)
  s->tuned.clear();
  for (auto &q : cand) s->tuned.push_back({q.lanes, q.ms});
  s->n_launches = 0;
  s->kernel_ms = 0;
  return AMWG_OK;
}

extern "C" {

int amwg_code_cache_stats(int64_t *hits, int64_t *misses, char *dir, size_t dir_capacity) {
  if (hits) *hits = g_cache_hits;
  if (misses) *misses = g_cache_misses;
  if (dir && dir_capacity) snprintf(dir, dir_capacity, "%s", cache_dir().c_str());
  return AMWG_OK;
}

int amwg_compile_user(const char *source, int32_t lanes_per_chain, int32_t block_threads, const char *arch, size_t *code_bytes) {
  if (!source || !arch) return fail(AMWG_EINVAL, "amwg_compile_user: null argument");
  std::vector<char> code;
  int rc = compile_user(source, lanes_per_chain, block_threads, arch, &code);
  if (rc == AMWG_OK && code_bytes) *code_bytes = code.size();
  return rc;
}

int amwg_create(const amwg_model_desc *m, const amwg_param_desc *params, int32_t n_params, const double *init,
                const amwg_comp_opt *comp_opts, const amwg_options *options, amwg_sampler **out) {
  if (!m || !params || !init || !comp_opts || !options || !out) return fail(AMWG_EINVAL, "amwg_create: null argument");
  if (n_params < 1 || n_params > kMaxIndex) return fail(AMWG_EINVAL, "amwg_create: %d named parameters (supported: 1..%d)", n_params, kMaxIndex);
  if (m->n_obs < 0) return fail(AMWG_EINVAL, "amwg_create: n_obs < 0");
  {
    int rc = check_options(options, model_max_threads(m->model));
    if (rc != AMWG_OK) return rc;
  }
  amwg_sampler *s = new amwg_sampler();
  s->opt = *options;
  s->model = m->model;
  s->C = options->chains;
  s->device = options->device;
  auto bail = [&](int rc) { amwg_destroy(s); return rc; };
#undef TRYB
#undef HIPB
#define TRYB(x) do { int rc_ = (x); if (rc_ != AMWG_OK) return bail(rc_); } while (0)
#define HIPB(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return bail(fail(AMWG_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_))); } while (0)
  PhaseClock clk;
  TRYB(build_layout(s, params, n_params, false));
  const int P = s->P;

  // ---- model / data checks
  const int N = m->n_obs;
  switch (m->model) {
    case AMWG_MODEL_NORMAL:
      if (P != 2 || n_params != 2) return bail(fail(AMWG_EINVAL, "normal model expects params {mu, sigma}"));
      if (!m->x && N) return bail(fail(AMWG_EINVAL, "normal model: x is null"));
      break;
    case AMWG_MODEL_BETA_BERN:
      if (P != 1) return bail(fail(AMWG_EINVAL, "beta_bern model expects params {theta}"));
      if (!m->x && N) return bail(fail(AMWG_EINVAL, "beta_bern model: x is null"));
      break;
    case AMWG_MODEL_HIER_NORMAL:
      if (n_params != 3 || m->G < 1 || m->G > 256 || params[0].len != m->G || P != m->G + 2)   // group ids are bytes in LDS
        return bail(fail(AMWG_EINVAL, "hier_normal model expects params {theta[G], mu, sigma}, 1 <= G <= 256 (more groups: write the closure, it is translated)"));
      if ((!m->x || !m->g) && N) return bail(fail(AMWG_EINVAL, "hier_normal model: y or g is null"));
      break;
    case AMWG_MODEL_POIS_GLM:
      if (n_params != 2 || params[0].len != 8 || P != 9 || m->K != 7)
        return bail(fail(AMWG_EINVAL, "pois_glm model expects params {beta[8], cp} and K = 7"));
      if ((!m->x || !m->y) && N) return bail(fail(AMWG_EINVAL, "pois_glm model: X or y is null"));
      if (N > (1 << 28)) return bail(fail(AMWG_EINVAL, "pois_glm model: %d observations (supported: up to 2^28; the kernel addresses a column with 32-bit byte offsets)", N));
      break;
    default: return bail(fail(AMWG_EINVAL, "unknown model id %d", m->model));
  }

  hipDeviceProp_t prop;
  clk.mark("layout + checks");
  TRYB(open_device(s, &prop));
  clk.mark("open device (HIP runtime)");
  // ---- model constants, with the kernel's own log (same roundings as the reference expression trees)
  ModelConsts &mc = s->mc;
  mc.neg_half_log_2pi = -0.5 * log_v8(2 * kPi);
  const double *h = m->hyper;
  if (m->model == AMWG_MODEL_NORMAL || m->model == AMWG_MODEL_HIER_NORMAL || m->model == AMWG_MODEL_POIS_GLM) {
    mc.m0 = h[0];
    mc.c0 = mc.neg_half_log_2pi - log_v8(h[1]);     // ld.norm's  -0.5*log(2*pi) - log(sd)
    mc.den0 = 2 * h[1] * h[1];                      //            (2*sd)*sd
  }
  if (m->model == AMWG_MODEL_NORMAL || m->model == AMWG_MODEL_HIER_NORMAL) {
    mc.ua = h[2]; mc.ub = h[3];
    mc.lunif = log_v8(1 / (h[3] - h[2]));           // ld.unif's log(1/(max-min))
  }
  if (m->model == AMWG_MODEL_HIER_NORMAL) {
    mc.c1 = mc.neg_half_log_2pi - log_v8(h[4]);
    mc.den1 = 2 * h[4] * h[4];
  }
  if (m->model == AMWG_MODEL_BETA_BERN) {
    mc.ba = h[0]; mc.bb = h[1];
    mc.lbeta_ab = lbeta_js(h[0], h[1]);
  }
  {   // reciprocals of the constant prior divisors, as the kernel's own make_reciprocal computes them
    const Reciprocal y0 = make_reciprocal(mc.den0), y1 = make_reciprocal(mc.den1);
    mc.y0_hi = y0.hi; mc.y0_lo = y0.lo; mc.den0_ok = (!options->exact_division && mid_range(mc.den0)) ? 1 : 0;
    mc.y1_hi = y1.hi; mc.y1_lo = y1.lo; mc.den1_ok = (!options->exact_division && mid_range(mc.den1)) ? 1 : 0;
  }
  mc.cp_upper = (double)(N - 1);
  mc.lunif_cp = log_v8(1 / (mc.cp_upper - 0.0));
  mc.exact_division = options->exact_division ? 1 : 0;
  mc.group_local = 0;
  mc.sufficient = 0;
  mc.suff_xbar_hi = mc.suff_xbar_lo = mc.suff_ss = 0.0;
  if (options->sufficient_statistics) {
    // amwg_options::sufficient_statistics: the two sufficient statistics of the Normal likelihood, in quad precision -- xbar as a double-double (its error must stay
    // far below an ulp of xbar - mu when mu sits next to the data: 2^-106 |xbar|), SS = sum (x_i - xbar)^2 rounded once
    if (m->model != AMWG_MODEL_NORMAL) return bail(fail(AMWG_EINVAL, "sufficient_statistics: only the Normal family has a pass-free certified value"));
    if (options->lanes_per_chain > 1) return bail(fail(AMWG_EINVAL, "sufficient_statistics decides from the one-lane certified kernel: lanes_per_chain must be 0 or 1"));
    __float128 sum = 0;
    for (int i = 0; i < N; ++i) sum += (__float128)m->x[i];
    const __float128 xbar = N > 0 ? sum / (__float128)N : (__float128)0;
    __float128 ss = 0;
    for (int i = 0; i < N; ++i) { const __float128 t = (__float128)m->x[i] - xbar; ss += t * t; }
    mc.suff_xbar_hi = (double)xbar;
    mc.suff_xbar_lo = (double)(xbar - (__float128)mc.suff_xbar_hi);
    mc.suff_ss = (double)ss;
    mc.sufficient = 1;
    s->opt.lanes_per_chain = 1;
  }
  GlLayoutHost gl;
  if (options->group_local) {
    // group-local evaluation (include/amwg.h, amwg_options::group_local; amwg_gl.h): the hierarchical family with a chain on one whole
    // wavefront, every lane serving one group -- which is what lets one pass evaluate all the proposals of a sweep over theta
    if (m->model != AMWG_MODEL_HIER_NORMAL) return bail(fail(AMWG_EINVAL, "group_local: only the hierarchical Normal family has a group-local evaluation"));
    if (n_params != 3 || !params[0].multidim || params[0].len != m->G || params[0].top != m->G)
      return bail(fail(AMWG_EINVAL, "group_local: parameters must be theta (dim [G]), mu, sigma"));
    if (options->lanes_per_chain != 0 && options->lanes_per_chain != 64) return bail(fail(AMWG_EINVAL, "group_local runs a chain on one wavefront: lanes_per_chain must be 0 or 64"));
    TRYB(gl_layout(m->x, m->g, N, m->G, &gl));
    s->opt.lanes_per_chain = 64;
    mc.group_local = 1;
  }

  // ---- data upload
  DataRef &d = s->d;
  d.n_obs = N; d.G = m->G; d.K = m->K;
  bool mid = true;
  if (m->model == AMWG_MODEL_NORMAL || m->model == AMWG_MODEL_HIER_NORMAL) {
    for (int i = 0; i < N; ++i) mid = mid && value_mid_range(m->x[i]);
    double *dx = nullptr;
    TRYB(dev_alloc(s, &dx, (size_t)N));
    if (N) HIPB(hipMemcpy(dx, m->x, (size_t)N * 8, hipMemcpyHostToDevice));
    d.x = dx;
    if (m->model == AMWG_MODEL_HIER_NORMAL) {
      std::vector<uint8_t> gb((size_t)N);
      for (int i = 0; i < N; ++i) {
        if (m->g[i] < 0 || m->g[i] >= m->G) return bail(fail(AMWG_EINVAL, "hier_normal: g[%d] = %d outside 0..%d", i, m->g[i], m->G - 1));
        gb[i] = (uint8_t)m->g[i];
      }
      uint8_t *dg = nullptr;
      TRYB(dev_alloc(s, &dg, (size_t)N));
      if (N) HIPB(hipMemcpy(dg, gb.data(), (size_t)N, hipMemcpyHostToDevice));
      d.xb = dg;
      if (mc.group_local) {      // the group-local kernel reads the lane-major tile and the lane table instead (amwg_gl.h)
        double *dt = nullptr;
        GlLane *dl = nullptr;
        TRYB(dev_alloc(s, &dt, gl.tile.size()));
        TRYB(dev_alloc(s, &dl, (size_t)64));
        HIPB(hipMemcpy(dt, gl.tile.data(), gl.tile.size() * 8, hipMemcpyHostToDevice));
        HIPB(hipMemcpy(dl, gl.lane, sizeof gl.lane, hipMemcpyHostToDevice));
        d.x = dt;
        d.arr[0] = dl;
        d.pad = gl.rounds;
        d.K = gl.n_min;
      }
    }
  } else if (m->model == AMWG_MODEL_BETA_BERN) {
    std::vector<uint8_t> xb((size_t)N);
    std::vector<uint32_t> xw(BetaBernModel::words(N), 0u), pre(BetaBernModel::words(N), 0u);
    bool invalid = false;
    for (int i = 0; i < N; ++i) {
      const bool one = m->x[i] == 1;
      invalid = invalid || !(one || m->x[i] == 0);
      xb[i] = one ? 1 : 0;
      if (one) xw[(size_t)i >> 5] |= 1u << (i & 31);
    }
    mc.has_invalid = invalid ? 1 : 0;
    const std::vector<uint32_t> tab = two_valued_tables(xb.data(), N);
    uint8_t *dxb = nullptr;
    uint32_t *dxw = nullptr, *dtab = nullptr;
    TRYB(dev_alloc(s, &dtab, tab.size()));
    HIPB(hipMemcpy(dtab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    d.arr[0] = dtab;
    TRYB(dev_alloc(s, &dxb, (size_t)N));
    TRYB(dev_alloc(s, &dxw, xw.size()));
    if (N) HIPB(hipMemcpy(dxb, xb.data(), (size_t)N, hipMemcpyHostToDevice));
    HIPB(hipMemcpy(dxw, xw.data(), xw.size() * 4, hipMemcpyHostToDevice));
    d.xb = dxb;
    d.xw = dxw;
  } else {  // POIS_GLM
    std::vector<double> lf((size_t)N);
    for (int i = 0; i < N; ++i) lf[i] = m->y[i] < 0 ? (double)INFINITY : lfactorial_js(m->y[i]);
    double *dX = nullptr, *dy = nullptr, *dlf = nullptr;
    TRYB(dev_alloc(s, &dX, (size_t)N * 7));
    TRYB(dev_alloc(s, &dy, (size_t)N));
    TRYB(dev_alloc(s, &dlf, (size_t)N));
    if (N) {
      std::vector<double> Xt((size_t)N * 7);   // row-major [N][7] -> column-major [7][N]
      for (int i = 0; i < N; ++i) for (int k = 0; k < 7; ++k) Xt[(size_t)k * N + i] = m->x[(size_t)i * 7 + k];
      HIPB(hipMemcpy(dX, Xt.data(), (size_t)N * 7 * 8, hipMemcpyHostToDevice));
      HIPB(hipMemcpy(dy, m->y, (size_t)N * 8, hipMemcpyHostToDevice));
      HIPB(hipMemcpy(dlf, lf.data(), (size_t)N * 8, hipMemcpyHostToDevice));
    }
    d.x = dX; d.y = dy; d.lfact = dlf;
    // what the bounds of the certified pass are made of (PoisGlmModel::log_post_approx)
    for (int k = 0; k < 7; ++k) { double mx = 0; for (int i = 0; i < N; ++i) { const double v = std::fabs(m->x[(size_t)i * 7 + k]); mx = (v > mx || v != v) ? v : mx; } mc.glm_xmax[k] = mx; }
    mc.glm_sum_y = 0; mc.glm_sum_lf = 0;
    for (int i = 0; i < N; ++i) { mc.glm_sum_y += std::fabs(m->y[i]); mc.glm_sum_lf += std::fabs(lf[i]); }
  }
  mc.data_mid_range = mid ? 1 : 0;
  clk.mark("device + data upload");

  TRYB(alloc_chain_state(s, params, n_params, init, comp_opts));
  clk.mark("chain state");

  // ---- geometry.  The constructor's warm-up log_post (mcmc.js:961-963) is folded into the first
  // launch (StepArgs.init_lp); amwg_chain_diag forces it with a 0-step launch if asked earlier.
  const size_t max_lds = prop.sharedMemPerBlock ? prop.sharedMemPerBlock : 65536;
  const int n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (m->model == AMWG_MODEL_HIER_NORMAL && !options->exact_division) {   // for which lane counts 2^j do the group labels repeat with the lane stride?
    for (int j = 0; j <= 10; ++j) {
      const int Gj = 1 << j;
      bool periodic = N > 0;
      for (int i = Gj; i < N && periodic; ++i) periodic = m->g[i] == m->g[i % Gj];
      if (periodic) s->hier_periodic_mask |= 1u << j;
    }
  }
  auto prepare = [&]() -> int {      // whatever depends on the lane count, once the geometry is fixed
    if (m->model == AMWG_MODEL_HIER_NORMAL) {   // HierNormalModel::pass_fast: constant-mean pass when the labels repeat with the lane stride
      int lg = 0;
      for (int g = s->lanes; g > 1; g >>= 1) ++lg;
      s->mc.group_lane_const = (int32_t)((s->hier_periodic_mask >> lg) & 1u);
      if (!s->mc.group_local)      // (the group-local kernel has its own use of DataRef::pad)
        s->d.pad = (hier_rows_wanted(s, s->lanes) && hier_rows_fit(s, s->block, max_lds)) ? HierNormalModel::row_pitch(s->d.n_obs) : 0;
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(s->kernel), hipFuncAttributeMaxDynamicSharedMemorySize, s->lds);
    return e == hipSuccess ? AMWG_OK : fail(AMWG_EHIP, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
  };
  if (options->lanes_per_chain == AMWG_LANES_AUTOTUNE) TRYB(autotune_geometry(s, n_cus, max_lds, prepare));
  else { TRYB(choose_geometry(s, n_cus, max_lds)); }
  clk.mark("geometry");
  TRYB(prepare());
  clk.mark("kernel attribute (module load)");
  HIPB(hipStreamSynchronize(s->stream));
  clk.mark("sync");
  *out = s;
  return AMWG_OK;
}

int amwg_create_user(const amwg_user_model *m, const amwg_param_desc *params, int32_t n_params, const double *init,
                     const amwg_comp_opt *comp_opts, const amwg_options *options, amwg_sampler **out) {
  if (!m || !m->source || !params || !init || !comp_opts || !options || !out) return fail(AMWG_EINVAL, "amwg_create_user: null argument");
  if (n_params < 1 || n_params > (1 << 20)) return fail(AMWG_EINVAL, "amwg_create_user: %d parameter entries (supported: 1..%d, of which at most %d stepped)", n_params, 1 << 20, kMaxIndex);
  if (m->n_arrays < 0) return fail(AMWG_EINVAL, "amwg_create_user: %d data arrays", m->n_arrays);
  if (m->n_arrays && (!m->arrays || !m->array_len)) return fail(AMWG_EINVAL, "amwg_create_user: arrays is null");
  if (m->n_derived < 0 || m->lds_bytes < 0) return fail(AMWG_EINVAL, "amwg_create_user: negative size");
  const int max_threads = m->max_threads > 0 ? (m->max_threads / 64) * 64 : 1024;
  if (max_threads < 64 || max_threads > 1024) return fail(AMWG_EINVAL, "amwg_create_user: max_threads must be in 64..1024");
  {
    int rc = check_options(options, max_threads);
    if (rc != AMWG_OK) return rc;
  }
  if (options->sufficient_statistics) return fail(AMWG_EINVAL, "sufficient_statistics: only the built-in Normal family has a pass-free certified value (translated closures: the certified tail's pass)");
  amwg_sampler *s = new amwg_sampler();
  s->opt = *options;
  s->model = 0;
  s->user = true;
  s->D = m->n_derived;
  s->user_lds = (m->lds_bytes + 15) & ~15;
  s->user_lds_one_lane = m->lds_bytes_one_lane > 0 ? ((m->lds_bytes_one_lane + 15) & ~15) : s->user_lds;
  s->user_parallel = m->parallel ? 1 : 0;
  s->user_max_threads = max_threads;
  s->user_work = m->work_per_eval;
  s->user_work_one_lane = m->work_one_lane;
  if (m->rows_n_obs < 0 || m->rows_groups < 0) { delete s; return fail(AMWG_EINVAL, "amwg_create_user: negative row plan"); }
  {
    // The row plan is honoured only as far as the GENERATED SOURCE states it (kRowN / kRowGroups / kRowSweep of translate.js): the source is what gets compiled, and
    // a caller built against an older amwg_user_model -- a shorter struct: the rows_* fields are then whatever follows it in memory -- must not switch a layout on
    // that the model has no code for (round-5 advisor finding).  The certified tail is read from the source alone (kCertifiedTail / kTailN): no struct field carries it.
    auto int_after = [&](const char *key) -> long {
      const char *q = strstr(m->source, key);
      return q ? strtol(q + strlen(key), nullptr, 10) : -1;
    };
    const long src_n = int_after("kRowN = "), src_g = int_after("kRowGroups = ");
    const bool src_sweep = strstr(m->source, "kRowSweep = true") != nullptr;
    const bool rows_ok = m->rows_n_obs > 0 && src_n == (long)m->rows_n_obs && src_g == (long)m->rows_groups;
    if (m->rows_n_obs > 0 && !rows_ok && src_n >= 0) { delete s; return fail(AMWG_EINVAL, "amwg_create_user: row plan (%d observations, %d groups) does not match the generated source (kRowN = %ld, kRowGroups = %ld)", m->rows_n_obs, m->rows_groups, src_n, src_g); }
    s->user_rows_n = rows_ok ? m->rows_n_obs : 0;
    s->user_rows_groups = rows_ok ? m->rows_groups : 0;
    s->user_rows_sweep = (rows_ok && m->rows_sweep && src_sweep) ? 1 : 0;
    s->user_rows_cert = s->user_rows_sweep && strstr(m->source, "kRowCert = true") != nullptr;
    const long tail_n = strstr(m->source, "kCertifiedTail = true") ? int_after("kTailN = ") : 0;
    s->user_cert_tail_n = tail_n > 0 && tail_n < (1l << 28) ? (int)tail_n : 0;
    const long ptail_n = strstr(m->source, "kPoisTail = true") ? int_after("kTailN = ") : 0;
    s->user_pois_tail_n = ptail_n > 0 && ptail_n < (1l << 28) ? (int)ptail_n : 0;
  }
  s->C = options->chains;
  s->device = options->device;
  auto bail = [&](int rc) { amwg_destroy(s); return rc; };
  TRYB(build_layout(s, params, n_params, true));
  for (int p = 0; p < n_params; ++p) s->user_has_binary = s->user_has_binary || params[p].type == AMWG_BINARY;
  hipDeviceProp_t prop;
  TRYB(open_device(s, &prop));

  // ---- data: every array the closure reads, as f64, row-major
  DataRef &d = s->d;
  d.n_obs = 0;
  std::vector<const void *> ext;      // arrays beyond the kInlineUserArrays pointers of the kernel arguments
  auto set_arr = [&](int j, const void *p) { if (j < kInlineUserArrays) d.arr[j] = p; else ext.push_back(p); };
  for (int j = 0; j < m->n_arrays; ++j) {
    const int64_t n = m->array_len[j];
    if (n < 0 || (n && !m->arrays[j])) return bail(fail(AMWG_EINVAL, "amwg_create_user: array %d is null or has a negative length", j));
    const int ty = m->array_type ? m->array_type[j] : AMWG_F64;
    if (ty == AMWG_F64) {
      double *p = nullptr;
      TRYB(dev_alloc(s, &p, (size_t)n));
      if (n) HIPB(hipMemcpy(p, m->arrays[j], (size_t)n * 8, hipMemcpyHostToDevice));
      set_arr(j, p);
    } else if (ty == AMWG_U8) {
      std::vector<uint8_t> tmp((size_t)n);
      for (int64_t i = 0; i < n; ++i) {
        const double v = m->arrays[j][i];
        if (!(v >= 0 && v <= 255 && v == (double)(uint8_t)v)) return bail(fail(AMWG_EINVAL, "amwg_create_user: array %d element %lld (%g) does not fit u8", j, (long long)i, v));
        tmp[(size_t)i] = (uint8_t)v;
      }
      uint8_t *p = nullptr;
      TRYB(dev_alloc(s, &p, (size_t)n + 16));
      if (n) HIPB(hipMemcpy(p, tmp.data(), (size_t)n, hipMemcpyHostToDevice));
      set_arr(j, p);
    } else if (ty == AMWG_I32) {
      std::vector<int32_t> tmp((size_t)n);
      for (int64_t i = 0; i < n; ++i) {
        const double v = m->arrays[j][i];
        if (!(v >= -2147483648.0 && v <= 2147483647.0 && v == (double)(int32_t)v)) return bail(fail(AMWG_EINVAL, "amwg_create_user: array %d element %lld (%g) does not fit i32", j, (long long)i, v));
        tmp[(size_t)i] = (int32_t)v;
      }
      int32_t *p = nullptr;
      TRYB(dev_alloc(s, &p, (size_t)n + 4));
      if (n) HIPB(hipMemcpy(p, tmp.data(), (size_t)n * 4, hipMemcpyHostToDevice));
      set_arr(j, p);
    } else {
      return bail(fail(AMWG_EINVAL, "amwg_create_user: array %d has unknown storage type %d", j, ty));
    }
  }
  if (!ext.empty()) {
    const void **d_ext = nullptr;
    TRYB(dev_alloc(s, &d_ext, ext.size()));
    HIPB(hipMemcpy(d_ext, ext.data(), ext.size() * sizeof(void *), hipMemcpyHostToDevice));
    d.arr_ext = d_ext;
  }
  TRYB(alloc_chain_state(s, params, n_params, init, comp_opts));

  const size_t max_lds = prop.sharedMemPerBlock ? prop.sharedMemPerBlock : 65536;
  const int n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  // ---- compile for the chosen geometry (cached per process by source text + geometry + arch) and load on this device
  auto prepare = [&]() -> int {
    static std::mutex mu;
    static std::map<std::string, std::vector<char>> cache;
    const std::string key = std::string(prop.gcnArchName) + "|" + std::to_string(s->lanes) + "|" + std::to_string(s->block) + "|" + m->source;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it == cache.end()) {
      std::vector<char> code;
      int rc = compile_user(m->source, s->lanes, s->block, prop.gcnArchName, &code);
      if (rc != AMWG_OK) return rc;
      it = cache.emplace(key, std::move(code)).first;
    }
    if (s->user_module) return AMWG_OK;      // (autotune hands back the module it kept)
    hipError_t e = hipModuleLoadData(&s->user_module, it->second.data());
    if (e == hipSuccess) e = hipModuleGetFunction(&s->user_fn, s->user_module, user_kernel_symbol(s));
    if (e != hipSuccess) {
      // the cache is never a requirement: an object the loader refuses (a planted or half-written file that still passed the checks, another
      // driver) is dropped and the closure compiled afresh, once
      (void)hipGetLastError();
      if (s->user_module) { (void)hipModuleUnload(s->user_module); s->user_module = nullptr; }
      std::vector<char> fresh;
      int rc = compile_user(m->source, s->lanes, s->block, prop.gcnArchName, &fresh, false);
      if (rc != AMWG_OK) return rc;
      it->second = std::move(fresh);
      e = hipModuleLoadData(&s->user_module, it->second.data());
      if (e == hipSuccess) e = hipModuleGetFunction(&s->user_fn, s->user_module, user_kernel_symbol(s));
      if (e != hipSuccess) return fail(AMWG_EHIP, "loading the compiled log_post failed: %s", hipGetErrorString(e));
    }
    // workgroups of this kernel use up to the whole 160 KB LDS of a CU; not every runtime needs (or accepts) the opt-in for module functions
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(s->user_fn), hipFuncAttributeMaxDynamicSharedMemorySize, s->lds);
    (void)hipGetLastError();
    return AMWG_OK;
  };
  if (options->lanes_per_chain == AMWG_LANES_AUTOTUNE) TRYB(autotune_geometry(s, n_cus, max_lds, prepare));
  else { TRYB(choose_geometry(s, n_cus, max_lds)); }
  TRYB(prepare());
  HIPB(hipStreamSynchronize(s->stream));
  *out = s;
  return AMWG_OK;
#undef TRYB
#undef HIPB
}

int amwg_destroy(amwg_sampler *s) {
  if (!s) return AMWG_OK;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  for (void *p : s->dev_allocs) (void)hipFree(p);
  if (s->d_draws) (void)hipFree(s->d_draws);
  if (s->user_module) (void)hipModuleUnload(s->user_module);
  for (hipEvent_t e : s->chunk_ev) (void)hipEventDestroy(e);
  if (s->copy_stream) (void)hipStreamDestroy(s->copy_stream);
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
  return AMWG_OK;
}

int amwg_burn_async(amwg_sampler *s, int64_t n) {
  if (!s || n < 0) return fail(AMWG_EINVAL, "amwg_burn: bad argument");
  HIP_TRY(hipSetDevice(s->device));
  return launch_steps(s, n, 1, nullptr);
}

int amwg_burn(amwg_sampler *s, int64_t n) {
  int rc = amwg_burn_async(s, n);
  if (rc != AMWG_OK) return rc;
  return finish_timing(s);
}

int amwg_sample_device(amwg_sampler *s, int64_t n, int64_t thin, double *out_dev, size_t out_bytes) {
  if (!s || n < 0 || thin < 1 || (!out_dev && n > 0)) return fail(AMWG_EINVAL, "amwg_sample_device: bad argument");
  const int64_t rows = (n + thin - 1) / thin;
  const size_t need = (size_t)rows * (size_t)(s->P + s->D) * (size_t)s->C * 8;
  if (out_bytes < need) return fail(AMWG_ESIZE, "amwg_sample: output needs %zu bytes, got %zu", need, out_bytes);
  HIP_TRY(hipSetDevice(s->device));
  int rc = launch_steps(s, n, thin, out_dev);
  if (rc != AMWG_OK) return rc;
  s->last_draws = out_dev;
  s->last_rows = rows;
  return AMWG_OK;
}

int amwg_sample_async(amwg_sampler *s, int64_t n, int64_t thin) {
  if (!s || n < 0 || thin < 1) return fail(AMWG_EINVAL, "amwg_sample: bad argument");
  const int64_t rows = (n + thin - 1) / thin;
  const size_t need = (size_t)rows * (size_t)(s->P + s->D) * (size_t)s->C * 8;
  HIP_TRY(hipSetDevice(s->device));
  if (need > s->d_draws_cap) {
    if (s->d_draws) {
      if (s->last_draws == s->d_draws) { s->last_draws = nullptr; s->last_rows = 0; }
      (void)hipFree(s->d_draws); s->d_draws = nullptr; s->d_draws_cap = 0;
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_draws), need ? need : 8));
    s->d_draws_cap = need;
  }
  if (!s->d_draws) {  // n == 0 before any allocation
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_draws), 8));
    s->d_draws_cap = 8;
  }
  return amwg_sample_device(s, n, thin, s->d_draws, need);
}

// Makes the pages of [p, p + bytes) resident without changing a byte.  A freshly allocated typed array / numpy array is untouched virtual memory, and a copy from the
// device into it runs at the speed the pages can be faulted in, not at the link's: measured on the GPU box (tools/ubench/pinned_copy.hip, 1 GiB) 9.8 GB/s into
// untouched pageable memory against 56 GB/s into the same memory once resident (pinning it first buys nothing more: 57 GB/s, and hipHostRegister / hipHostMalloc
// of a gigabyte cost 55-170 ms themselves).  Round 5 touched the pages with ONE thread -- ~6 GB/s, slower than the kernels produce rows at cfg2 (10 GB/s): sample()
// took three times its kernels' time.  Now: transparent huge pages are asked for (512 times fewer faults where the host grants them), the kernel is asked to populate
// the range in one call (MADV_POPULATE_WRITE, Linux 5.14), and where that is not available the pages are touched -- a write of the value just read.
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static void prefault(char *p, size_t bytes) {
  if (!p || !bytes) return;
  const size_t page = 4096;
  const uintptr_t a0 = ((uintptr_t)p + page - 1) & ~(uintptr_t)(page - 1), a1 = ((uintptr_t)p + bytes) & ~(uintptr_t)(page - 1);
  if (a1 > a0 && madvise(reinterpret_cast<void *>(a0), a1 - a0, MADV_POPULATE_WRITE) == 0) {
    volatile char *q = p;
    q[0] = q[0];
    q[bytes - 1] = q[bytes - 1];      // (the partial pages at either end)
    return;
  }
  volatile char *q = p;
  for (size_t o = 0; o < bytes; o += page) q[o] = q[o];
  q[bytes - 1] = q[bytes - 1];
}
// ... by a few helper threads that run AHEAD of the copies: the rows of a sample call leave the device launch by launch (below), and the destination of launch j's rows must
// be resident when its kernel ends.  The helpers walk the destination in the order the copies will (chunk by chunk, slice by slice) and publish how far they are.
namespace {
struct Prefaulter {
  struct Piece { char *p; size_t bytes; size_t chunk; };
  std::vector<Piece> pieces;              // in copy order
  std::vector<std::thread> workers;
  std::atomic<size_t> next{0};
  std::vector<std::atomic<int>> done;     // pieces finished per chunk
  std::vector<int> per_chunk;
  explicit Prefaulter(size_t n_chunks) : done(n_chunks), per_chunk(n_chunks, 0) { for (auto &d : done) d.store(0); }
  void add(char *p, size_t bytes, size_t chunk) {
    const size_t step = (size_t)8 << 20;      // 8 MB pieces: several helpers share one chunk's range
    for (size_t o = 0; o < bytes; o += step) { pieces.push_back({p + o, bytes - o < step ? bytes - o : step, chunk}); per_chunk[chunk]++; }
  }
  void start(int n_threads) {
    for (int t = 0; t < n_threads; ++t)
      workers.emplace_back([this] {
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= pieces.size()) return;
          prefault(pieces[i].p, pieces[i].bytes);
          done[pieces[i].chunk].fetch_add(1, std::memory_order_release);
        }
      });
  }
  void wait_chunk(size_t j) {      // (the caller helps instead of idling: it takes pieces too)
    while (done[j].load(std::memory_order_acquire) < per_chunk[j]) {
      const size_t i = next.fetch_add(1);
      if (i < pieces.size()) { prefault(pieces[i].p, pieces[i].bytes); done[pieces[i].chunk].fetch_add(1, std::memory_order_release); }
      else std::this_thread::yield();
    }
  }
  ~Prefaulter() { for (auto &w : workers) w.join(); }
};
}  // namespace

// The rows of a sample call are final launch by launch (launch_steps records an event after each): the rows of launch j leave the
// device on copy_stream while launches j + 1, ... run on the sampler's stream -- a pageable destination (a JavaScript typed array, a numpy
// array) makes each copy block THIS thread, not the GPU.  65 536 chains x 1000 draws x 2 components are 1.05 GB: with the destination resident
// in time (Prefaulter) they leave at the link's rate behind the kernels that produce them.
int amwg_fetch_draws_slices(amwg_sampler *s, int32_t n_slices, const int32_t *base, const int32_t *len, double *const *out, const size_t *out_bytes) {
  if (!s) return fail(AMWG_EINVAL, "amwg_fetch_draws: null sampler");
  if (s->last_draws != s->d_draws) return fail(AMWG_EINVAL, "amwg_fetch_draws: no amwg_sample_async pending");
  if (n_slices < 0 || (n_slices > 0 && (!base || !len || !out || !out_bytes))) return fail(AMWG_EINVAL, "amwg_fetch_draws_slices: bad argument");
  const int PR = s->P + s->D;
  const size_t C = (size_t)s->C;
  size_t total_bytes = 0;
  for (int k = 0; k < n_slices; ++k) {
    if (base[k] < 0 || len[k] < 0 || base[k] > PR || len[k] > PR - base[k]) return fail(AMWG_EINVAL, "amwg_fetch_draws_slices: slice %d = [%d, %d) outside the %d recorded values", k, base[k], base[k] + len[k], PR);
    const size_t need = (size_t)s->last_rows * (size_t)len[k] * C * 8;
    if (need && !out[k]) return fail(AMWG_EINVAL, "amwg_fetch_draws: null output");
    if (out_bytes[k] < need) return fail(AMWG_ESIZE, "amwg_sample: output needs %zu bytes, got %zu", need, out_bytes[k]);
    total_bytes += need;
  }
  HIP_TRY(hipSetDevice(s->device));
  if (!s->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
  // (launches since the sample call -- a burn in between -- have reset the per-launch marks: then everything is final once the stream is idle)
  const bool marks = !s->chunk_rows.empty() && s->chunk_rows.back() == s->last_rows;
  const size_t n_chunks = marks ? s->chunk_rows.size() : 1;
  // the destination becomes resident ahead of the copies, in their order.  Helpers only where there is something to win (>= 16 MB): small results are touched inline
  Prefaulter pf(n_chunks);
  {
    int64_t r0 = 0;
    for (size_t j = 0; j < n_chunks; ++j) {
      const int64_t r1 = marks ? s->chunk_rows[j] : s->last_rows;
      if (r1 > r0)
        for (int k = 0; k < n_slices; ++k) {
          if (!len[k]) continue;
          const size_t width = (size_t)len[k] * C * 8;
          pf.add(reinterpret_cast<char *>(out[k]) + (size_t)r0 * width, (size_t)(r1 - r0) * width, j);
        }
      r0 = r1;
    }
    for (int k = 0; k < n_slices; ++k) {      // transparent huge pages for the whole destination, where the host grants them on request
      const size_t need = (size_t)s->last_rows * (size_t)len[k] * C * 8;
      const uintptr_t a0 = ((uintptr_t)out[k] + 4095) & ~(uintptr_t)4095, a1 = ((uintptr_t)out[k] + need) & ~(uintptr_t)4095;
      if (need >= ((size_t)4 << 20) && a1 > a0) (void)madvise(reinterpret_cast<void *>(a0), a1 - a0, MADV_HUGEPAGE);
    }
    unsigned hw = std::thread::hardware_concurrency();
    int helpers = total_bytes >= ((size_t)16 << 20) ? (hw >= 8 ? 4 : (hw >= 4 ? 2 : (hw >= 2 ? 1 : 0))) : 0;
    if (const char *e = getenv("AMWG_PREFAULT_THREADS")) helpers = atoi(e) < 0 ? 0 : (atoi(e) > 16 ? 16 : atoi(e));
    pf.start(helpers);
  }
  if (!marks) HIP_TRY(hipStreamSynchronize(s->stream));
  int64_t r0 = 0;
  for (size_t j = 0; j < n_chunks; ++j) {
    const int64_t r1 = marks ? s->chunk_rows[j] : s->last_rows;
    if (r1 > r0) {
      pf.wait_chunk(j);      // (while launch j still runs, usually: the helpers are ahead)
      if (marks) HIP_TRY(hipEventSynchronize(s->chunk_ev[j]));
      for (int k = 0; k < n_slices; ++k) {
        if (!len[k]) continue;
        const size_t width = (size_t)len[k] * C * 8, spitch = (size_t)PR * C * 8;
        const char *src = reinterpret_cast<const char *>(s->d_draws) + (size_t)r0 * spitch + (size_t)base[k] * C * 8;
        char *dst = reinterpret_cast<char *>(out[k]) + (size_t)r0 * width;
        if (len[k] == PR) HIP_TRY(hipMemcpyAsync(dst, src, (size_t)(r1 - r0) * width, hipMemcpyDeviceToHost, s->copy_stream));
        else HIP_TRY(hipMemcpy2DAsync(dst, width, src, spitch, width, (size_t)(r1 - r0), hipMemcpyDeviceToHost, s->copy_stream));
      }
      HIP_TRY(hipStreamSynchronize(s->copy_stream));
    }
    r0 = r1;
  }
  return finish_timing(s);      // (kernel_ms / launch_info of the LATEST call on the stream, whichever path was taken)
}

int amwg_fetch_draws(amwg_sampler *s, double *out, size_t out_bytes) {
  if (!s) return fail(AMWG_EINVAL, "amwg_fetch_draws: null sampler");
  const int32_t base = 0, len = s->P + s->D;
  return amwg_fetch_draws_slices(s, 1, &base, &len, &out, &out_bytes);
}

int amwg_sample(amwg_sampler *s, int64_t n, int64_t thin, double *out, size_t out_bytes) {
  if (!s || n < 0 || thin < 1 || (!out && n > 0)) return fail(AMWG_EINVAL, "amwg_sample: bad argument");
  const int64_t rows = (n + thin - 1) / thin;
  const size_t need = (size_t)rows * (size_t)(s->P + s->D) * (size_t)s->C * 8;
  if (out_bytes < need) return fail(AMWG_ESIZE, "amwg_sample: output needs %zu bytes, got %zu", need, out_bytes);
  int rc = amwg_sample_async(s, n, thin);
  if (rc != AMWG_OK) return rc;
  return amwg_fetch_draws(s, out, out_bytes);
}

int amwg_sync(amwg_sampler *s) {
  if (!s) return fail(AMWG_EINVAL, "amwg_sync: null sampler");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return finish_timing(s);
}

int amwg_set_adapting(amwg_sampler *s, int32_t flag) {
  if (!s) return fail(AMWG_EINVAL, "amwg_set_adapting: null sampler");
  HIP_TRY(hipSetDevice(s->device));
  for (auto &b : s->h_adapt) b = flag ? 1 : 0;
  HIP_TRY(hipMemcpyAsync(s->d_adapt, s->h_adapt.data(), s->h_adapt.size(), hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return AMWG_OK;
}

int amwg_get_state(amwg_sampler *s, double *out, size_t out_bytes) {
  if (!s || !out) return fail(AMWG_EINVAL, "amwg_get_state: null argument");
  const size_t need = (size_t)s->P * (size_t)s->C * 8;
  if (out_bytes < need) return fail(AMWG_ESIZE, "amwg_get_state: output needs %zu bytes, got %zu", need, out_bytes);
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipMemcpyAsync(out, s->ch.state, need, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return AMWG_OK;
}

int amwg_set_state(amwg_sampler *s, const double *state, size_t state_bytes) {
  if (!s || !state) return fail(AMWG_EINVAL, "amwg_set_state: null argument");
  const size_t need = (size_t)s->P * (size_t)s->C * 8;
  if (state_bytes != need) return fail(AMWG_ESIZE, "amwg_set_state: expected %zu bytes, got %zu", need, state_bytes);
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  HIP_TRY(hipMemcpy(s->ch.state, state, need, hipMemcpyHostToDevice));
  s->lp_ready = false;   // the next launch recomputes log_post(state) first
  return AMWG_OK;
}

int amwg_last_sample_diagnostics(amwg_sampler *s, double *rhat, double *ess) {
  if (!s || !rhat || !ess) return fail(AMWG_EINVAL, "amwg_last_sample_diagnostics: null argument");
  if (!s->last_draws || s->last_rows < 4 || s->C < 2) return fail(AMWG_EINVAL, "amwg_last_sample_diagnostics: needs a sample() of >= 4 kept draws on >= 2 chains");
  HIP_TRY(hipSetDevice(s->device));
  const int PR = s->P + s->D;
  const size_t C = (size_t)s->C, n_out = 4 * (size_t)PR * C;
  DevBuf dout;
  HIP_TRY(dout.alloc(n_out * 8));
  hipLaunchKernelGGL(chain_halves_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)PR), dim3(256), 0, s->stream, s->last_draws, s->last_rows, PR, s->C, dout.as<double>());
  std::vector<double> h(n_out);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(h.data(), dout.p, n_out * 8, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  const double n = (double)(s->last_rows / 2), m = 2.0 * (double)C;   // 2C half-chains of n draws
  for (int p = 0; p < PR; ++p) {
    // W = mean within-sequence variance; B/n = variance of the sequence means (over the 2C halves)
    long double sw = 0, sm = 0;
    for (int hf = 0; hf < 2; ++hf)
      for (size_t c = 0; c < C; ++c) { sm += h[((size_t)(hf * 2 + 0) * PR + p) * C + c]; sw += h[((size_t)(hf * 2 + 1) * PR + p) * C + c]; }
    const double W = (double)(sw / m), gm = (double)(sm / m);
    long double sb = 0, sbc = 0;
    for (int hf = 0; hf < 2; ++hf)
      for (size_t c = 0; c < C; ++c) { const double dlt = h[((size_t)(hf * 2 + 0) * PR + p) * C + c] - gm; sb += (long double)dlt * dlt; }
    const double B_over_n = (double)(sb / (m - 1));
    const double var_plus = (n - 1) / n * W + B_over_n;
    rhat[p] = W > 0 ? std::sqrt(var_plus / W) : (double)NAN;
    // whole-chain means: average of the two half means (equal lengths)
    long double smc = 0;
    for (size_t c = 0; c < C; ++c) smc += 0.5 * (h[((size_t)0 * PR + p) * C + c] + h[((size_t)2 * PR + p) * C + c]);
    const double gmc = (double)(smc / (double)C);
    for (size_t c = 0; c < C; ++c) { const double dlt = 0.5 * (h[((size_t)0 * PR + p) * C + c] + h[((size_t)2 * PR + p) * C + c]) - gmc; sbc += (long double)dlt * dlt; }
    const double var_chain_mean = (double)(sbc / ((double)C - 1));
    ess[p] = var_chain_mean > 0 ? (double)C * var_plus / var_chain_mean : (double)NAN;
  }
  return AMWG_OK;
}

int amwg_info(amwg_sampler *s, double *pls, int32_t *ac, int32_t *it, int32_t *bc, int64_t *acc, int64_t *inb) {
  if (!s) return fail(AMWG_EINVAL, "amwg_info: null sampler");
  const size_t PC = (size_t)s->P * (size_t)s->C;
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (pls) HIP_TRY(hipMemcpy(pls, s->ch.prop_log_scale, PC * 8, hipMemcpyDeviceToHost));
  if (ac) HIP_TRY(hipMemcpy(ac, s->ch.acceptance_count, PC * 4, hipMemcpyDeviceToHost));
  if (it) HIP_TRY(hipMemcpy(it, s->ch.iterations_since_adaption, PC * 4, hipMemcpyDeviceToHost));
  if (bc) HIP_TRY(hipMemcpy(bc, s->ch.batch_count, PC * 4, hipMemcpyDeviceToHost));
  std::vector<int32_t> tmp;
  if (acc) {
    tmp.resize(PC);
    HIP_TRY(hipMemcpy(tmp.data(), s->ch.accepts, PC * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < PC; ++i) acc[i] = tmp[i];
  }
  if (inb) {
    tmp.resize(PC);
    HIP_TRY(hipMemcpy(tmp.data(), s->ch.inbounds, PC * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < PC; ++i) inb[i] = tmp[i];
  }
  return AMWG_OK;
}

int amwg_chain_diag(amwg_sampler *s, uint64_t *uniforms, double *log_post_out, int32_t *named_order) {
  if (!s) return fail(AMWG_EINVAL, "amwg_chain_diag: null sampler");
  const size_t C = (size_t)s->C;
  HIP_TRY(hipSetDevice(s->device));
  // (log_post of the current state as the expression gives it: a 0-step launch computes it where it was never formed or where the stepper's cheaper value stands in)
  if (!s->lp_ready || !s->lp_is_expression) { int rc = launch_steps(s, 0, 1, nullptr, true); if (rc != AMWG_OK) return rc; }
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (uniforms) HIP_TRY(hipMemcpy(uniforms, s->ch.rng_n, C * 8, hipMemcpyDeviceToHost));
  if (log_post_out) HIP_TRY(hipMemcpy(log_post_out, s->ch.lp_curr, C * 8, hipMemcpyDeviceToHost));
  if (named_order && s->ch.perm16) {
    std::vector<uint16_t> p16((size_t)s->n_params * C);
    HIP_TRY(hipMemcpy(p16.data(), s->ch.perm16, p16.size() * 2, hipMemcpyDeviceToHost));
    for (size_t c = 0; c < C; ++c)
      for (int k = 0; k < s->n_params; ++k) named_order[c * s->n_params + k] = (int32_t)p16[(size_t)k * C + c];
  } else if (named_order) {
    std::vector<uint64_t> pv(C);
    HIP_TRY(hipMemcpy(pv.data(), s->ch.perm, C * 8, hipMemcpyDeviceToHost));
    for (size_t c = 0; c < C; ++c)
      for (int k = 0; k < s->n_params; ++k) named_order[c * s->n_params + k] = (int32_t)((pv[c] >> (4 * k)) & 0xF);
  }
  return AMWG_OK;
}

int amwg_last_sample_moments(amwg_sampler *s, double *mean, double *sd) {
  if (!s || !mean || !sd) return fail(AMWG_EINVAL, "amwg_last_sample_moments: null argument");
  if (!s->last_draws || s->last_rows < 1) return fail(AMWG_EINVAL, "amwg_last_sample_moments: no sample() call yet");
  HIP_TRY(hipSetDevice(s->device));
  DevBuf buf;
  const int PR = s->P + s->D;
  HIP_TRY(buf.alloc((size_t)PR * 16));
  double *dm = buf.as<double>();
  hipLaunchKernelGGL(moments_kernel, dim3(PR), dim3(1024), 0, s->stream, s->last_draws, s->last_rows, PR, s->C, dm, dm + PR);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(mean, dm, (size_t)PR * 8, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipMemcpyAsync(sd, dm + PR, (size_t)PR * 8, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return AMWG_OK;
}

int amwg_tuning(const amwg_sampler *s, int32_t *lanes, double *ms, int32_t cap) {
  if (!s) return 0;
  for (int32_t i = 0; i < cap && i < (int32_t)s->tuned.size(); ++i) {
    if (lanes) lanes[i] = s->tuned[i].first;
    if (ms) ms[i] = s->tuned[i].second;
  }
  return (int)s->tuned.size();
}

int amwg_num_components(const amwg_sampler *s) { return s ? s->P : 0; }
int amwg_num_recorded(const amwg_sampler *s) { return s ? s->P + s->D : 0; }
int64_t amwg_num_chains(const amwg_sampler *s) { return s ? s->C : 0; }

int amwg_launch_info(const amwg_sampler *s, int32_t *lanes, int32_t *block, int32_t *grid, int32_t *lds, int32_t *n_launches, double *kernel_ms) {
  if (!s) return fail(AMWG_EINVAL, "amwg_launch_info: null sampler");
  if (lanes) *lanes = s->lanes;
  if (block) *block = s->block;
  if (grid) *grid = s->grid;
  if (lds) *lds = s->lds;
  if (n_launches) *n_launches = s->n_launches;
  if (kernel_ms) *kernel_ms = s->kernel_ms;
  return AMWG_OK;
}

int amwg_summation_order(const amwg_sampler *s) {
  if (!s) return fail(AMWG_EINVAL, "amwg_summation_order: null sampler");
  // (the certified kernels of the Poisson and the hierarchical family evaluate the expression in the reference's order: amwg_kernel.h kRefOrder)
  if (s->lanes > 1 && certified_kernel(s) && (s->user || s->model == AMWG_MODEL_POIS_GLM || s->model == AMWG_MODEL_HIER_NORMAL)) return 1;      // (a closure: amwg_user_sweep_cert)
  return s->lanes;
}

const char *amwg_kernel_name(const amwg_sampler *s) {
  if (!s) { (void)fail(AMWG_EINVAL, "amwg_kernel_name: null sampler"); return ""; }
  amwg_sampler *m = const_cast<amwg_sampler *>(s);
  if (m->kernel_name.empty()) {
    const int cls = s->block <= 256 ? 256 : (s->block <= 512 ? 512 : 1024);
    char buf[96];
    if (s->user) snprintf(buf, sizeof buf, "%s", user_kernel_symbol(s));
    else if (s->mc.group_local) snprintf(buf, sizeof buf, "amwg_gl_kernel<HierGlModel,%d>", cls);
    else if (s->model == AMWG_MODEL_HIER_NORMAL && s->d.pad > 0) snprintf(buf, sizeof buf, "amwg_sweep_kernel%s<HierNormalModel,%d>", s->certified ? "_cert" : "", cls);
    else {
      static const char *const fam[] = {"NormalModel", "BetaBernModel", "HierNormalModel", "PoisGlmModel"};
      const int f = s->model == AMWG_MODEL_NORMAL ? 0 : (s->model == AMWG_MODEL_BETA_BERN ? 1 : (s->model == AMWG_MODEL_HIER_NORMAL ? 2 : 3));
      snprintf(buf, sizeof buf, "amwg_step_kernel%s<%s,%d,%d>", s->certified ? "_cert" : "", fam[f], s->lanes, s->lanes > 64 ? (s->lanes <= 256 ? 256 : (s->lanes <= 512 ? 512 : 1024)) : cls);
    }
    m->kernel_name = buf;
  }
  return m->kernel_name.c_str();
}

int amwg_fp64_peak(int32_t device, double *lane_ops_per_s) {
  if (!lane_ops_per_s) return fail(AMWG_EINVAL, "amwg_fp64_peak: null argument");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev < 1) return fail(AMWG_EHIP, "no HIP device available (%s)", hipGetErrorString(e));
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  const int blocks = (prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256) * 2, threads = 1024, iters = 20000;
  DevBuf dout;
  HIP_TRY(dout.alloc((size_t)blocks * threads * 8));
  EventPair ev;
  HIP_TRY(hipEventCreate(&ev.e0));
  HIP_TRY(hipEventCreate(&ev.e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {    // first repetition warms the clocks up
    HIP_TRY(hipEventRecord(ev.e0, 0));
    hipLaunchKernelGGL(fp64_peak_kernel, dim3(blocks), dim3(threads), 0, 0, dout.as<double>(), iters, 0.999999, 1e-7);
    HIP_TRY(hipEventRecord(ev.e1, 0));
    HIP_TRY(hipEventSynchronize(ev.e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, ev.e0, ev.e1));
    if (rep > 0 && ms < best) best = ms;
  }
  *lane_ops_per_s = (double)blocks * threads * (double)iters * 64.0 / (best * 1e-3);
  return AMWG_OK;
}

#if defined(AMWG_AUDIT) || defined(AMWG_X_PHASES)
// include/amwg_selftest.h: what the audited launches of this sampler have recorded so far (and optionally a reset)
int amwg_audit_fetch(amwg_sampler *s, double *per_chain, uint64_t *hist, int32_t reset) {
  if (!s) return fail(AMWG_EINVAL, "amwg_audit_fetch: null sampler");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (per_chain) HIP_TRY(hipMemcpy(per_chain, s->ch.audit, (size_t)4 * s->C * 8, hipMemcpyDeviceToHost));
  if (hist) HIP_TRY(hipMemcpy(hist, s->ch.audit_hist, 128 * 8, hipMemcpyDeviceToHost));
  if (reset) { HIP_TRY(hipMemset(s->ch.audit, 0, (size_t)4 * s->C * 8)); HIP_TRY(hipMemset(s->ch.audit_hist, 0, 128 * 8)); }
  return AMWG_OK;
}
#endif

#if defined(AMWG_SELFTEST)
// include/amwg_selftest.h: the host-side machinery that makes sample()'s destination resident ahead of the device-to-host copies (Prefaulter above), run on a
// caller's buffer cut into `n_chunks` chunks with `threads` helpers: no byte may change, whatever the alignment and the sizes.  No GPU involved.
int amwg_prefault_selftest(char *buf, size_t bytes, int32_t n_chunks, int32_t threads) {
  if (!buf || n_chunks < 1 || threads < 0 || threads > 16) return fail(AMWG_EINVAL, "amwg_prefault_selftest: bad argument");
  Prefaulter pf((size_t)n_chunks);
  const size_t per = bytes / (size_t)n_chunks;
  for (int32_t j = 0; j < n_chunks; ++j) pf.add(buf + (size_t)j * per, j == n_chunks - 1 ? bytes - (size_t)j * per : per, (size_t)j);
  pf.start(threads);
  for (int32_t j = 0; j < n_chunks; ++j) pf.wait_chunk((size_t)j);
  return AMWG_OK;
}

int amwg_two_valued_sum_check(int32_t device, const double *x, int32_t n, int64_t m, const double *acc0, const double *l1, const double *l0,
                              double *out_fast_forward, double *out_term_by_term) {
  if (!x || !acc0 || !l1 || !l0 || !out_fast_forward || !out_term_by_term || n < 0 || m < 0) return fail(AMWG_EINVAL, "amwg_two_valued_sum_check: bad argument");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev < 1) return fail(AMWG_EHIP, "no HIP device available (%s)", hipGetErrorString(e));
  HIP_TRY(hipSetDevice(device));
  std::vector<uint8_t> xb((size_t)n);
  for (int i = 0; i < n; ++i) xb[i] = x[i] == 1 ? 1 : 0;
  const std::vector<uint32_t> tab = two_valued_tables(xb.data(), n);
  DevBuf dtab, d[5];
  HIP_TRY(dtab.alloc(tab.size() * 4));
  HIP_TRY(hipMemcpy(dtab.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  const double *src[3] = {acc0, l1, l0};
  for (int k = 0; k < 5; ++k) {
    HIP_TRY(d[k].alloc((size_t)m * 8));
    if (k < 3 && m) HIP_TRY(hipMemcpy(d[k].p, src[k], (size_t)m * 8, hipMemcpyHostToDevice));
  }
  if (m) hipLaunchKernelGGL(two_valued_check_kernel, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, 0, dtab.as<uint32_t>(), n, m,
                            d[0].as<double>(), d[1].as<double>(), d[2].as<double>(), d[3].as<double>(), d[4].as<double>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out_fast_forward, d[3].p, (size_t)m * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_term_by_term, d[4].p, (size_t)m * 8, hipMemcpyDeviceToHost));
  return AMWG_OK;
}

int amwg_ld_device(int32_t device, int64_t n, const double *records, double *out) {
  if (!records || !out || n < 0) return fail(AMWG_EINVAL, "amwg_ld_device: bad argument");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev < 1) return fail(AMWG_EHIP, "no HIP device available (%s)", hipGetErrorString(e));
  HIP_TRY(hipSetDevice(device));
  DevBuf dr, dout;
  HIP_TRY(dr.alloc((size_t)n * 40));
  HIP_TRY(dout.alloc((size_t)n * 8));
  HIP_TRY(hipMemcpy(dr.p, records, (size_t)n * 40, hipMemcpyHostToDevice));
  if (n) hipLaunchKernelGGL(amwg_ld_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, n, dr.as<double>(), dout.as<double>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dout.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  return AMWG_OK;
}

int amwg_device_eval(int32_t device, int32_t op, int64_t n, const double *a, const double *b, const double *c, double *out) {
  if (!a || !out || n < 0) return fail(AMWG_EINVAL, "amwg_device_eval: bad argument");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev < 1) return fail(AMWG_EHIP, "no HIP device available (%s)", hipGetErrorString(e));
  HIP_TRY(hipSetDevice(device));
  DevBuf da, db, dc, dout;
  const size_t bytes = (size_t)n * 8;
  HIP_TRY(da.alloc(bytes));
  HIP_TRY(dout.alloc(bytes));
  HIP_TRY(hipMemcpy(da.p, a, bytes, hipMemcpyHostToDevice));
  if (b) { HIP_TRY(db.alloc(bytes)); HIP_TRY(hipMemcpy(db.p, b, bytes, hipMemcpyHostToDevice)); }
  if (c) { HIP_TRY(dc.alloc(bytes)); HIP_TRY(hipMemcpy(dc.p, c, bytes, hipMemcpyHostToDevice)); }
  if (n) hipLaunchKernelGGL(amwg_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, op, n, da.as<double>(), db.as<double>(), dc.as<double>(), dout.as<double>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
  return AMWG_OK;
}

#endif   // AMWG_SELFTEST

}  // extern "C"
