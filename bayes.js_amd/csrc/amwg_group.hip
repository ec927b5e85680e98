// amwg_group.hip -- posterior summaries over SEVERAL samplers that are the shards of one logical job (chains split over the
// devices of one node by amwg_options.chain_offset).  Each device reduces its own draws; the per-device
// partial results (a few doubles per recorded component) are combined with an RCCL all-reduce over xGMI -- the one
// collective of this path besides the gather of raw draws ("all-gather of per-chain moment summaries").
//
//   * one process drives all devices: ncclCommInitAll over the shards' devices, collective calls inside ncclGroupStart/End,
//     each on its shard's own stream; the communicator is cached per device list for the life of the process;
//   * RCCL is loaded on first use (dlopen librccl.so.1): single-device users never pay for it, and a missing library is a
//     loud AMWG_EHIP, not a fallback;
//   * shards that share a device (tests on a one-GPU box; oversubscription) are first summed into that device's leader
//     shard by a kernel -- RCCL refuses duplicate devices in a communicator -- then the leaders all-reduce, then the result
//     is copied back to the followers.  A one-device group still runs the (one-rank) RCCL all-reduce.
//
// Statistics (same definitions as the single-sampler entry points in amwg_core.hip):
//   moments      two passes: all-reduce of (count, sum) -> mean; all-reduce of sum (x - mean)^2 -> sd (n - 1)
//   diagnostics  split-R-hat / ESS from per-chain half means and variances (chain_halves): all-reduce of
//                (sum var_h, sum mean_h, sum chain_mean) -> W, grand means; all-reduce of (sum (mean_h - gm)^2, sum (cm - gmc)^2)
//   quantiles    grouped ncclSend/ncclRecv of every shard's values of one component to the first shard's device, radix sort there
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "amwg_sampler.h"

namespace {

// ---- RCCL, loaded on first use -------------------------------------------------------------------------------------------
struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  // one process per device (amwg_comm_*): a communicator from a shared id, and what it says about itself
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  std::string error;
  Rccl() {
    for (const char *name : {"librccl.so.1", "librccl.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) { error = std::string("cannot load librccl.so (") + dlerror() + ")"; return; }
    auto sym = [&](const char *n) { void *p = dlsym(lib, n); if (!p && error.empty()) error = std::string("librccl.so lacks ") + n; return p; };
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
    AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
    Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
    Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
    CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    CommCount = reinterpret_cast<decltype(CommCount)>(sym("ncclCommCount"));
    CommUserRank = reinterpret_cast<decltype(CommUserRank)>(sym("ncclCommUserRank"));
    CommCuDevice = reinterpret_cast<decltype(CommCuDevice)>(sym("ncclCommCuDevice"));
    AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
  }
};
Rccl &rccl() { static Rccl r; return r; }

#define HIPG(expr)                                                                                                        \
  do {                                                                                                                    \
    hipError_t e_ = (expr);                                                                                               \
    if (e_ != hipSuccess) return amwg_fail(AMWG_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_));                       \
  } while (0)
#define NCCLG(expr)                                                                                                       \
  do {                                                                                                                    \
    ncclResult_t r_ = (expr);                                                                                             \
    if (r_ != ncclSuccess) return amwg_fail(AMWG_EHIP, "%s failed: %s", #expr, rccl().GetErrorString(r_));                  \
  } while (0)

// One group = the distinct devices of the shards (leaders) with a communicator over them, plus a scratch buffer per shard.
struct Group {
  std::vector<amwg_sampler *> shards;
  std::vector<int> leader_of;          // shard -> index of the first shard on the same device
  std::vector<int> leaders;            // shard indices that lead a device, in shard order (= RCCL ranks)
  std::vector<int> rank_of;            // shard -> rank of its leader
  std::vector<ncclComm_t> comms;       // per leader
  std::vector<double *> buf;           // per shard: device scratch (vector being reduced)
  size_t buf_len = 0;
  ~Group() {
    for (size_t i = 0; i < buf.size(); ++i)
      if (buf[i]) { (void)hipSetDevice(shards[i]->device); (void)hipFree(buf[i]); }
  }
};

int get_comms(const std::vector<int> &devs, std::vector<ncclComm_t> *out) {
  static std::mutex mu;
  static std::map<std::vector<int>, std::vector<ncclComm_t>> cache;      // kept for the life of the process
  Rccl &R = rccl();
  if (!R.error.empty()) return amwg_fail(AMWG_EHIP, "multi-device summaries need RCCL: %s", R.error.c_str());
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(devs);
  if (it == cache.end()) {
    std::vector<ncclComm_t> comms(devs.size());
    NCCLG(R.CommInitAll(comms.data(), (int)devs.size(), devs.data()));
    it = cache.emplace(devs, std::move(comms)).first;
  }
  *out = it->second;
  return AMWG_OK;
}

// An error return between ncclGroupStart and ncclGroupEnd would leave the process-wide RCCL group open -- and the communicators are cached for
// the life of the process, so every later collective would misbehave or hang.  The guard closes the group on any exit path.
struct RcclGroupGuard {
  Rccl &R;
  bool open = false;
  explicit RcclGroupGuard(Rccl &r) : R(r) {}
  ncclResult_t start() { const ncclResult_t rc = R.GroupStart(); open = rc == ncclSuccess; return rc; }
  ncclResult_t end() { open = false; return R.GroupEnd(); }
  ~RcclGroupGuard() { if (open) (void)R.GroupEnd(); }
};

int open_group(amwg_sampler *const *shards, int n, size_t buf_len, bool need_draws, Group *g) {
  if (!shards || n < 1) return amwg_fail(AMWG_EINVAL, "amwg_group: no samplers");
  const int PR = shards[0]->P + shards[0]->D;
  for (int i = 0; i < n; ++i) {
    amwg_sampler *s = shards[i];
    if (!s) return amwg_fail(AMWG_EINVAL, "amwg_group: sampler %d is null", i);
    if (s->P + s->D != PR) return amwg_fail(AMWG_EINVAL, "amwg_group: sampler %d records %d values per draw, sampler 0 %d", i, s->P + s->D, PR);
    if (need_draws && (!s->last_draws || s->last_rows < 1)) return amwg_fail(AMWG_EINVAL, "amwg_group: sampler %d has no sample() call yet", i);
    if (need_draws && s->last_rows != shards[0]->last_rows) return amwg_fail(AMWG_EINVAL, "amwg_group: sampler %d kept %lld draws, sampler 0 %lld", i, (long long)s->last_rows, (long long)shards[0]->last_rows);
    g->shards.push_back(s);
  }
  g->leader_of.assign(n, -1);
  g->rank_of.assign(n, -1);
  std::vector<int> devs;
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < i; ++j) if (shards[j]->device == shards[i]->device) { g->leader_of[i] = g->leader_of[j]; break; }
    if (g->leader_of[i] < 0) { g->leader_of[i] = i; g->rank_of[i] = (int)g->leaders.size(); g->leaders.push_back(i); devs.push_back(shards[i]->device); }
    else g->rank_of[i] = g->rank_of[g->leader_of[i]];
  }
  int rc = get_comms(devs, &g->comms);
  if (rc != AMWG_OK) return rc;
  g->buf.assign(n, nullptr);
  g->buf_len = buf_len;
  for (int i = 0; i < n; ++i) {
    HIPG(hipSetDevice(shards[i]->device));
    HIPG(hipMalloc(reinterpret_cast<void **>(&g->buf[i]), (buf_len ? buf_len : 1) * 8));
  }
  return AMWG_OK;
}

// device memory freed on every exit path
struct DevOwned {
  void *p = nullptr;
  ~DevOwned() { if (p) (void)hipFree(p); }
};

__global__ void add_into_kernel(double *dst, const double *src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

// sum of buf[i][0..count) over all shards, left in EVERY shard's buf (and copied to `host` from shard 0)
int all_reduce_sum(Group &g, int count, double *host) {
  Rccl &R = rccl();
  const int n = (int)g.shards.size();
  // followers -> their device's leader (same device: the leader's stream waits for the follower's partial result first)
  for (int i = 0; i < n; ++i) {
    const int L = g.leader_of[i];
    if (L == i) continue;
    HIPG(hipSetDevice(g.shards[i]->device));
    HIPG(hipStreamSynchronize(g.shards[i]->stream));
    hipLaunchKernelGGL(add_into_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, g.shards[L]->stream, g.buf[L], g.buf[i], count);
    HIPG(hipGetLastError());
  }
  // leaders: RCCL all-reduce (one rank per distinct device; in place)
  {
    RcclGroupGuard grp(R);
    NCCLG(grp.start());
    for (size_t r = 0; r < g.leaders.size(); ++r) {
      const int i = g.leaders[r];
      HIPG(hipSetDevice(g.shards[i]->device));
      NCCLG(R.AllReduce(g.buf[i], g.buf[i], (size_t)count, ncclDouble, ncclSum, g.comms[r], g.shards[i]->stream));
    }
    NCCLG(grp.end());
  }
  for (size_t r = 0; r < g.leaders.size(); ++r) {
    const int i = g.leaders[r];
    HIPG(hipSetDevice(g.shards[i]->device));
    HIPG(hipStreamSynchronize(g.shards[i]->stream));
  }
  // leaders -> followers
  for (int i = 0; i < n; ++i) {
    const int L = g.leader_of[i];
    if (L == i) continue;
    HIPG(hipSetDevice(g.shards[i]->device));
    HIPG(hipMemcpyAsync(g.buf[i], g.buf[L], (size_t)count * 8, hipMemcpyDeviceToDevice, g.shards[i]->stream));
    HIPG(hipStreamSynchronize(g.shards[i]->stream));
  }
  if (host) {
    HIPG(hipSetDevice(g.shards[0]->device));
    HIPG(hipMemcpy(host, g.buf[0], (size_t)count * 8, hipMemcpyDeviceToHost));
  }
  return AMWG_OK;
}

// ---- per-shard reductions ---------------------------------------------------------------------------------------------------
// block p: out[p] = sum over the shard's recorded draws of f(x), f = x (center == nullptr) or (x - center[p])^2
__global__ void __launch_bounds__(1024) draw_sums_kernel(const double *draws, int64_t rows, int PR, int64_t C, const double *center, double *out) {
  __shared__ double red[1024];
  const int p = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int64_t n = rows * C;
  const double m = center ? center[p] : 0.0;
  double sum = 0;
  for (int64_t i = tid; i < n; i += nt) {
    const double x = draws[((i / C) * PR + p) * C + (i % C)];
    sum += center ? (x - m) * (x - m) : x;
  }
  red[tid] = sum;
  __syncthreads();
  for (int o = nt / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  if (tid == 0) out[p] = red[0];
}

// per chain and recorded value: mean and (n-1) variance of each half of the chain's kept draws (the same kernel as in
// amwg_core.hip): out[((h*2 + stat) * PR + p) * C + c]
__global__ void group_chain_halves_kernel(const double *draws, int64_t rows, int PR, int64_t C, double *out) {
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

// block p.  stage 0: out[p] = sum of half variances, out[PR + p] = sum of half means, out[2 PR + p] = sum of chain means
//           stage 1: out[p] = sum (half mean - gm[p])^2, out[PR + p] = sum (chain mean - gm[PR + p])^2
__global__ void __launch_bounds__(256) halves_sums_kernel(const double *hv, int PR, int64_t C, int stage, const double *gm, double *out) {
  __shared__ double red[3][256];
  const int p = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  double a = 0, b = 0, c3 = 0;
  for (int64_t c = tid; c < C; c += nt) {
    const double m0 = hv[((size_t)0 * PR + p) * C + c], v0 = hv[((size_t)1 * PR + p) * C + c];
    const double m1 = hv[((size_t)2 * PR + p) * C + c], v1 = hv[((size_t)3 * PR + p) * C + c];
    const double cm = 0.5 * (m0 + m1);
    if (stage == 0) { a += v0 + v1; b += m0 + m1; c3 += cm; }
    else { const double g0 = gm[p], g1 = gm[PR + p]; a += (m0 - g0) * (m0 - g0) + (m1 - g0) * (m1 - g0); b += (cm - g1) * (cm - g1); }
  }
  red[0][tid] = a; red[1][tid] = b; red[2][tid] = c3;
  __syncthreads();
  for (int o = nt / 2; o > 0; o >>= 1) {
    if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; red[2][tid] += red[2][tid + o]; }
    __syncthreads();
  }
  if (tid == 0) { out[p] = red[0][0]; out[PR + p] = red[1][0]; if (stage == 0) out[2 * PR + p] = red[2][0]; }
}

__global__ void gather_component_kernel2(const double *draws, int64_t rows, int PR, int64_t C, int p, double *out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C) return;
  out[i] = draws[((i / C) * PR + p) * C + (i % C)];
}

__global__ void pick_quantiles_kernel2(const double *sorted, int64_t n, const double *probs, int n_probs, double *out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_probs) return;
  const double q = probs[k];
  if (!(q >= 0.0 && q <= 1.0)) { out[k] = __builtin_nan(""); return; }
  const double h = (double)(n - 1) * q;
  const int64_t lo = (int64_t)floor(h);
  const int64_t hi = lo + 1 < n ? lo + 1 : lo;
  out[k] = sorted[lo] + (h - (double)lo) * (sorted[hi] - sorted[lo]);
}

}  // namespace

extern "C" {

int amwg_group_moments(amwg_sampler *const *shards, int32_t n, double *mean, double *sd) {
  if (!mean || !sd) return amwg_fail(AMWG_EINVAL, "amwg_group_moments: null argument");
  Group g;
  const int PR0 = (shards && n > 0 && shards[0]) ? shards[0]->P + shards[0]->D : 0;
  int rc = open_group(shards, n, (size_t)PR0 + 1, true, &g);
  if (rc != AMWG_OK) return rc;
  const int PR = PR0;
  std::vector<double> h((size_t)PR + 1);
  // pass 1: (sum per component, count)
  for (int i = 0; i < n; ++i) {
    amwg_sampler *s = g.shards[i];
    HIPG(hipSetDevice(s->device));
    hipLaunchKernelGGL(draw_sums_kernel, dim3(PR), dim3(1024), 0, s->stream, s->last_draws, s->last_rows, PR, s->C, (const double *)nullptr, g.buf[i]);
    HIPG(hipGetLastError());
    const double cnt = (double)s->last_rows * (double)s->C;
    HIPG(hipMemcpyAsync(g.buf[i] + PR, &cnt, 8, hipMemcpyHostToDevice, s->stream));
    HIPG(hipStreamSynchronize(s->stream));   // `cnt` lives on this stack frame
  }
  rc = all_reduce_sum(g, PR + 1, h.data());
  if (rc != AMWG_OK) return rc;
  const double N = h[PR];
  for (int p = 0; p < PR; ++p) mean[p] = h[p] / N;
  // pass 2: sum of squared deviations from the global mean
  std::vector<double *> centers(n, nullptr);
  struct FreeAll { std::vector<double *> &v; std::vector<amwg_sampler *> &s; ~FreeAll() { for (size_t i = 0; i < v.size(); ++i) if (v[i]) { (void)hipSetDevice(s[i]->device); (void)hipFree(v[i]); } } } free_all{centers, g.shards};
  for (int i = 0; i < n; ++i) {
    amwg_sampler *s = g.shards[i];
    HIPG(hipSetDevice(s->device));
    HIPG(hipMalloc(reinterpret_cast<void **>(&centers[i]), (size_t)PR * 8));
    HIPG(hipMemcpyAsync(centers[i], mean, (size_t)PR * 8, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(draw_sums_kernel, dim3(PR), dim3(1024), 0, s->stream, s->last_draws, s->last_rows, PR, s->C, (const double *)centers[i], g.buf[i]);
    HIPG(hipGetLastError());
  }
  rc = all_reduce_sum(g, PR, h.data());
  if (rc != AMWG_OK) return rc;
  for (int p = 0; p < PR; ++p) sd[p] = N > 1 ? std::sqrt(h[p] / (N - 1)) : 0.0;
  return AMWG_OK;
}

int amwg_group_diagnostics(amwg_sampler *const *shards, int32_t n, double *rhat, double *ess) {
  if (!rhat || !ess) return amwg_fail(AMWG_EINVAL, "amwg_group_diagnostics: null argument");
  Group g;
  const int PR0 = (shards && n > 0 && shards[0]) ? shards[0]->P + shards[0]->D : 0;
  int rc = open_group(shards, n, (size_t)3 * PR0 + 1, true, &g);
  if (rc != AMWG_OK) return rc;
  const int PR = PR0;
  int64_t Ctot = 0;
  for (int i = 0; i < n; ++i) Ctot += g.shards[i]->C;
  if (g.shards[0]->last_rows < 4 || Ctot < 2) return amwg_fail(AMWG_EINVAL, "amwg_group_diagnostics: needs a sample() of >= 4 kept draws on >= 2 chains");
  std::vector<double *> halves(n, nullptr), gmd(n, nullptr);
  struct FreeAll { std::vector<double *> &a, &b; std::vector<amwg_sampler *> &s; ~FreeAll() { for (size_t i = 0; i < a.size(); ++i) { (void)hipSetDevice(s[i]->device); if (a[i]) (void)hipFree(a[i]); if (b[i]) (void)hipFree(b[i]); } } } free_all{halves, gmd, g.shards};
  for (int i = 0; i < n; ++i) {
    amwg_sampler *s = g.shards[i];
    const size_t C = (size_t)s->C;
    HIPG(hipSetDevice(s->device));
    HIPG(hipMalloc(reinterpret_cast<void **>(&halves[i]), 4 * (size_t)PR * C * 8));
    HIPG(hipMalloc(reinterpret_cast<void **>(&gmd[i]), 2 * (size_t)PR * 8));
    hipLaunchKernelGGL(group_chain_halves_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)PR), dim3(256), 0, s->stream, s->last_draws, s->last_rows, PR, s->C, halves[i]);
    HIPG(hipGetLastError());
    hipLaunchKernelGGL(halves_sums_kernel, dim3(PR), dim3(256), 0, s->stream, (const double *)halves[i], PR, s->C, 0, (const double *)nullptr, g.buf[i]);
    HIPG(hipGetLastError());
  }
  std::vector<double> h((size_t)3 * PR);
  rc = all_reduce_sum(g, 3 * PR, h.data());
  if (rc != AMWG_OK) return rc;
  const double nh = (double)(g.shards[0]->last_rows / 2), m = 2.0 * (double)Ctot;      // 2C half-chains of nh draws
  std::vector<double> W(PR), gm(2 * (size_t)PR);
  for (int p = 0; p < PR; ++p) { W[p] = h[p] / m; gm[p] = h[PR + p] / m; gm[PR + p] = h[2 * PR + p] / (double)Ctot; }
  for (int i = 0; i < n; ++i) {
    amwg_sampler *s = g.shards[i];
    HIPG(hipSetDevice(s->device));
    HIPG(hipMemcpyAsync(gmd[i], gm.data(), 2 * (size_t)PR * 8, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(halves_sums_kernel, dim3(PR), dim3(256), 0, s->stream, (const double *)halves[i], PR, s->C, 1, (const double *)gmd[i], g.buf[i]);
    HIPG(hipGetLastError());
  }
  rc = all_reduce_sum(g, 2 * PR, h.data());
  if (rc != AMWG_OK) return rc;
  for (int p = 0; p < PR; ++p) {
    const double B_over_n = h[p] / (m - 1);
    const double var_plus = (nh - 1) / nh * W[p] + B_over_n;
    rhat[p] = W[p] > 0 ? std::sqrt(var_plus / W[p]) : (double)NAN;
    const double var_chain_mean = h[PR + p] / ((double)Ctot - 1);
    ess[p] = var_chain_mean > 0 ? (double)Ctot * var_plus / var_chain_mean : (double)NAN;
  }
  return AMWG_OK;
}

int amwg_group_quantiles(amwg_sampler *const *shards, int32_t n, const double *probs, int32_t n_probs, double *out) {
  if (!probs || !out || n_probs < 1) return amwg_fail(AMWG_EINVAL, "amwg_group_quantiles: bad argument");
  Group g;
  int rc = open_group(shards, n, 1, true, &g);
  if (rc != AMWG_OK) return rc;
  Rccl &R = rccl();
  const int PR = g.shards[0]->P + g.shards[0]->D;
  std::vector<int64_t> cnt(n), off(n);
  int64_t total = 0;
  for (int i = 0; i < n; ++i) { cnt[i] = g.shards[i]->last_rows * g.shards[i]->C; off[i] = total; total += cnt[i]; }
  if (total > 2147483647) return amwg_fail(AMWG_EINVAL, "amwg_group_quantiles: more than 2^31 values per component");
  amwg_sampler *root = g.shards[0];
  // root: all values of one component, sorted copy, radix-sort scratch; other shards: their own values of the component
  std::vector<double *> vals(n, nullptr);
  double *all = nullptr, *sorted = nullptr, *dprobs = nullptr, *dout = nullptr;
  void *tmp = nullptr;
  struct FreeAll {
    std::vector<double *> &v; std::vector<amwg_sampler *> &s; double *&a, *&b, *&c, *&d; void *&t;
    ~FreeAll() {
      for (size_t i = 1; i < v.size(); ++i) if (v[i]) { (void)hipSetDevice(s[i]->device); (void)hipFree(v[i]); }
      (void)hipSetDevice(s[0]->device);
      (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); (void)hipFree(d); (void)hipFree(t);
    }
  } free_all{vals, g.shards, all, sorted, dprobs, dout, tmp};
  HIPG(hipSetDevice(root->device));
  HIPG(hipMalloc(reinterpret_cast<void **>(&all), (size_t)total * 8));
  HIPG(hipMalloc(reinterpret_cast<void **>(&sorted), (size_t)total * 8));
  HIPG(hipMalloc(reinterpret_cast<void **>(&dprobs), (size_t)n_probs * 8));
  HIPG(hipMalloc(reinterpret_cast<void **>(&dout), (size_t)n_probs * 8));
  HIPG(hipMemcpyAsync(dprobs, probs, (size_t)n_probs * 8, hipMemcpyHostToDevice, root->stream));
  size_t tmp_bytes = 0;
  HIPG(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, all, sorted, (int)total, 0, 64, root->stream));
  HIPG(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 8));
  vals[0] = all;
  for (int i = 1; i < n; ++i) {
    if (g.shards[i]->device == root->device) { vals[i] = nullptr; continue; }      // same device: gathers straight into `all`
    HIPG(hipSetDevice(g.shards[i]->device));
    HIPG(hipMalloc(reinterpret_cast<void **>(&vals[i]), (size_t)cnt[i] * 8));
  }
  for (int p = 0; p < PR; ++p) {
    for (int i = 0; i < n; ++i) {
      amwg_sampler *s = g.shards[i];
      const bool local = s->device == root->device;
      double *dst = local ? all + off[i] : vals[i];
      HIPG(hipSetDevice(s->device));
      hipLaunchKernelGGL(gather_component_kernel2, dim3((unsigned)((cnt[i] + 255) / 256)), dim3(256), 0, s->stream, s->last_draws, s->last_rows, PR, s->C, p, dst);
      HIPG(hipGetLastError());
      if (local && i != 0) HIPG(hipStreamSynchronize(s->stream));     // another stream of the root's device wrote into `all`
    }
    // remote shards: RCCL point-to-point to the root's rank (grouped: the sends and receives progress together)
    bool any_remote = false;
    for (int i = 1; i < n; ++i) any_remote = any_remote || g.shards[i]->device != root->device;
    if (any_remote) {
      for (int i = 1; i < n; ++i)      // (constraints are checked BEFORE the group is opened)
        if (g.shards[i]->device != root->device && g.leader_of[i] != i)
          return amwg_fail(AMWG_EINVAL, "amwg_group_quantiles: two shards share a device other than the first shard's");
      RcclGroupGuard grp(R);
      NCCLG(grp.start());
      for (int i = 1; i < n; ++i) {
        amwg_sampler *s = g.shards[i];
        if (s->device == root->device) continue;
        HIPG(hipSetDevice(s->device));
        NCCLG(R.Send(vals[i], (size_t)cnt[i], ncclDouble, 0, g.comms[g.rank_of[i]], s->stream));
        HIPG(hipSetDevice(root->device));
        NCCLG(R.Recv(all + off[i], (size_t)cnt[i], ncclDouble, g.rank_of[i], g.comms[0], root->stream));
      }
      NCCLG(grp.end());
    }
    HIPG(hipSetDevice(root->device));
    HIPG(hipcub::DeviceRadixSort::SortKeys(tmp, tmp_bytes, all, sorted, (int)total, 0, 64, root->stream));
    hipLaunchKernelGGL(pick_quantiles_kernel2, dim3((unsigned)((n_probs + 63) / 64)), dim3(64), 0, root->stream, sorted, total, dprobs, n_probs, dout);
    HIPG(hipGetLastError());
    HIPG(hipMemcpyAsync(out + (size_t)p * n_probs, dout, (size_t)n_probs * 8, hipMemcpyDeviceToHost, root->stream));
    HIPG(hipStreamSynchronize(root->stream));
    for (int i = 1; i < n; ++i) { HIPG(hipSetDevice(g.shards[i]->device)); HIPG(hipStreamSynchronize(g.shards[i]->stream)); }
  }
  return AMWG_OK;
}

// ---- the gather at sample collection (north_star: "RCCL-over-xGMI gather only at sample collection") -------------------
// ONE process, one sampler per device: every shard's block of recorded draws [rows][PR][C_i] travels to the device of shard `root`
// (grouped ncclSend / ncclRecv over the communicator of the shards' devices; shards on the root's own device are copied), where the
// blocks stand back to back in shard order -- then, if asked, ONE copy to the host.  offsets[i] (optional) = first element of shard i's
// block.  The Node front-end copies each shard straight to the host instead by default (eight PCIe links in parallel beat funnelling
// everything through one GPU when the destination is host memory); this is for a caller who wants the job's draws
// in ONE device buffer, and what bench.py --inproc times as `gather`.
int amwg_group_gather_draws(amwg_sampler *const *shards, int32_t n, int32_t root_index, double *dst_device, double *dst_host, size_t capacity_bytes, int64_t *offsets) {
  if (root_index < 0 || root_index >= n) return amwg_fail(AMWG_EINVAL, "amwg_group_gather_draws: root %d outside 0..%d", root_index, n - 1);
  Group g;
  int rc = open_group(shards, n, 1, true, &g);
  if (rc != AMWG_OK) return rc;
  Rccl &R = rccl();
  const int PR = g.shards[0]->P + g.shards[0]->D;
  std::vector<int64_t> cnt(n), off(n);
  int64_t total = 0;
  for (int i = 0; i < n; ++i) { cnt[i] = g.shards[i]->last_rows * PR * g.shards[i]->C; off[i] = total; total += cnt[i]; if (offsets) offsets[i] = off[i]; }
  if ((size_t)total * 8 > capacity_bytes) return amwg_fail(AMWG_ESIZE, "amwg_group_gather_draws: %lld bytes needed, %zu given", (long long)total * 8, capacity_bytes);
  amwg_sampler *root = g.shards[root_index];
  // the shards' own streams have the draws in flight: wait for them, and hear what their step kernels had to say (amwg_sync reads the device
  // error word: a launch that refused itself must not be gathered as if it had produced draws) -- a caller need not have called amwg_sync
  for (int i = 0; i < n; ++i) {
    rc = amwg_sync(g.shards[i]);
    if (rc != AMWG_OK) return rc;
  }
  for (int i = 0; i < n; ++i)      // (constraints are checked BEFORE the RCCL group is opened)
    if (g.shards[i]->device != root->device && g.leader_of[i] != i)
      return amwg_fail(AMWG_EINVAL, "amwg_group_gather_draws: two shards share a device other than the root's");
  DevOwned all;
  double *dst = dst_device;
  HIPG(hipSetDevice(root->device));
  if (!dst) { HIPG(hipMalloc(&all.p, (size_t)(total ? total : 1) * 8)); dst = static_cast<double *>(all.p); }
  for (int i = 0; i < n; ++i) {
    amwg_sampler *s = g.shards[i];
    if (s->device != root->device) continue;
    HIPG(hipSetDevice(s->device));
    HIPG(hipMemcpyAsync(dst + off[i], s->last_draws, (size_t)cnt[i] * 8, hipMemcpyDeviceToDevice, root->stream));
  }
  bool any_remote = false;
  for (int i = 0; i < n; ++i) any_remote = any_remote || g.shards[i]->device != root->device;
  if (any_remote) {
    RcclGroupGuard grp(R);
    NCCLG(grp.start());
    for (int i = 0; i < n; ++i) {
      amwg_sampler *s = g.shards[i];
      if (s->device == root->device) continue;
      HIPG(hipSetDevice(s->device));
      NCCLG(R.Send(s->last_draws, (size_t)cnt[i], ncclDouble, g.rank_of[root_index], g.comms[g.rank_of[i]], s->stream));
      HIPG(hipSetDevice(root->device));
      NCCLG(R.Recv(dst + off[i], (size_t)cnt[i], ncclDouble, g.rank_of[i], g.comms[g.rank_of[root_index]], root->stream));
    }
    NCCLG(grp.end());
  }
  for (int i = 0; i < n; ++i) { HIPG(hipSetDevice(g.shards[i]->device)); HIPG(hipStreamSynchronize(g.shards[i]->stream)); }
  HIPG(hipSetDevice(root->device));
  HIPG(hipStreamSynchronize(root->stream));
  if (dst_host) HIPG(hipMemcpy(dst_host, dst, (size_t)total * 8, hipMemcpyDeviceToHost));
  return AMWG_OK;
}

// what the communicator of a group of shards says about itself: ranks (ncclCommCount) and the device of every rank
int amwg_group_comm_info(amwg_sampler *const *shards, int32_t n, int32_t *n_ranks, int32_t *devices, int32_t capacity) {
  Group g;
  int rc = open_group(shards, n, 1, false, &g);
  if (rc != AMWG_OK) return rc;
  Rccl &R = rccl();
  int count = 0;
  NCCLG(R.CommCount(g.comms[0], &count));
  if (n_ranks) *n_ranks = count;
  for (size_t r = 0; r < g.comms.size() && devices && (int32_t)r < capacity; ++r) {
    int dev = -1;
    NCCLG(R.CommCuDevice(g.comms[r], &dev));
    devices[r] = dev;
  }
  return AMWG_OK;
}

// ---- one PROCESS per device (torch.distributed.run, MPI, ...): the same two exchanges over a communicator built from a shared id ---------------
// amwg_comm_unique_id on one rank -> the 128 bytes travel to the others by whatever the host has (a file, a socket, torch's store) ->
// amwg_comm_create on every rank (collective) -> amwg_comm_gather_draws / amwg_comm_moments (collective) -> amwg_comm_destroy.
struct amwg_comm {
  ncclComm_t comm = nullptr;
  int n_ranks = 0, rank = 0, device = 0;
};

int amwg_comm_unique_id(char *id, size_t capacity) {
  Rccl &R = rccl();
  if (!R.error.empty()) return amwg_fail(AMWG_EHIP, "amwg_comm: RCCL is not available: %s", R.error.c_str());
  if (!id || capacity < sizeof(ncclUniqueId)) return amwg_fail(AMWG_EINVAL, "amwg_comm_unique_id: the id needs %zu bytes", sizeof(ncclUniqueId));
  ncclUniqueId u;
  NCCLG(R.GetUniqueId(&u));
  memcpy(id, &u, sizeof u);
  return AMWG_OK;
}

int amwg_comm_create(const char *id, size_t id_bytes, int32_t n_ranks, int32_t rank, int32_t device, amwg_comm **out) {
  Rccl &R = rccl();
  if (!R.error.empty()) return amwg_fail(AMWG_EHIP, "amwg_comm: RCCL is not available: %s", R.error.c_str());
  if (!id || !out || id_bytes < sizeof(ncclUniqueId) || n_ranks < 1 || rank < 0 || rank >= n_ranks) return amwg_fail(AMWG_EINVAL, "amwg_comm_create: bad argument");
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  HIPG(hipSetDevice(device));
  amwg_comm *c = new amwg_comm();
  c->n_ranks = n_ranks; c->rank = rank; c->device = device;
  const ncclResult_t r = R.CommInitRank(&c->comm, n_ranks, u, rank);
  if (r != ncclSuccess) { delete c; return amwg_fail(AMWG_EHIP, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, n_ranks, device, R.GetErrorString(r)); }
  *out = c;
  return AMWG_OK;
}

int amwg_comm_info(amwg_comm *c, int32_t *n_ranks, int32_t *rank, int32_t *device) {
  if (!c) return amwg_fail(AMWG_EINVAL, "amwg_comm_info: null communicator");
  Rccl &R = rccl();
  int count = 0, me = -1, dev = -1;
  NCCLG(R.CommCount(c->comm, &count));      // what RCCL itself reports, not what the caller said
  NCCLG(R.CommUserRank(c->comm, &me));
  NCCLG(R.CommCuDevice(c->comm, &dev));
  if (n_ranks) *n_ranks = count;
  if (rank) *rank = me;
  if (device) *device = dev;
  return AMWG_OK;
}

int amwg_comm_destroy(amwg_comm *c) {
  if (!c) return AMWG_OK;
  Rccl &R = rccl();
  if (c->comm) (void)R.CommDestroy(c->comm);
  delete c;
  return AMWG_OK;
}

// Collective: the block of recorded draws [rows][PR][C] of this rank's last sample call travels to rank `root`, where the blocks stand
// back to back in rank order in dst_device (root only; capacity checked there); counts (optional, n_ranks entries, every rank) = the
// elements each rank contributed.  The blocks may differ in size (uneven shards): the counts are exchanged first (an 8-byte all-gather).
int amwg_comm_gather_draws(amwg_sampler *s, amwg_comm *c, int32_t root, double *dst_device, size_t capacity_bytes, int64_t *counts) {
  if (!s || !c) return amwg_fail(AMWG_EINVAL, "amwg_comm_gather_draws: null argument");      // (nothing to take part in a collective WITH)
  // Conditions only THIS rank can see are not returned before the collective -- the other ranks would sit in it for good -- but travel with it:
  // a rank that cannot contribute all-gathers a count of -1, and every rank then returns an error without any send or receive posted.
  char why[200] = "";
  if (root < 0 || root >= c->n_ranks) snprintf(why, sizeof why, "root %d outside 0..%d", root, c->n_ranks - 1);
  else if (!s->last_draws || s->last_rows < 1) snprintf(why, sizeof why, "no sample() call yet");
  else if (s->device != c->device) snprintf(why, sizeof why, "the sampler is on device %d, the communicator on %d", s->device, c->device);
  else if (amwg_sync(s) != AMWG_OK) snprintf(why, sizeof why, "%.190s", amwg_last_error());      // (incl. the step kernels' device error word)
  const bool bad = why[0] != 0;
  Rccl &R = rccl();
  HIPG(hipSetDevice(c->device));
  hipStream_t st = (s->device == c->device) ? s->stream : nullptr;
  const int64_t mine = bad ? -1 : s->last_rows * (int64_t)(s->P + s->D) * s->C;
  DevOwned cnt_dev;
  HIPG(hipMalloc(&cnt_dev.p, (size_t)(c->n_ranks + 1) * 8));
  int64_t *d_all = static_cast<int64_t *>(cnt_dev.p), *d_mine = d_all + c->n_ranks;
  HIPG(hipMemcpyAsync(d_mine, &mine, 8, hipMemcpyHostToDevice, st));
  NCCLG(R.AllGather(d_mine, d_all, 1, ncclInt64, c->comm, st));
  std::vector<int64_t> cnt((size_t)c->n_ranks);
  HIPG(hipMemcpyAsync(cnt.data(), d_all, (size_t)c->n_ranks * 8, hipMemcpyDeviceToHost, st));
  HIPG(hipStreamSynchronize(st));
  if (bad) return amwg_fail(AMWG_EINVAL, "amwg_comm_gather_draws: %s (every rank was told; nothing was exchanged)", why);
  for (int r = 0; r < c->n_ranks; ++r)
    if (cnt[r] < 0) return amwg_fail(AMWG_EINVAL, "amwg_comm_gather_draws: rank %d could not contribute (its own call says why); nothing was exchanged", r);
  if (counts) for (int r = 0; r < c->n_ranks; ++r) counts[r] = cnt[r];
  if (c->rank == root) {
    int64_t total = 0;
    for (int r = 0; r < c->n_ranks; ++r) total += cnt[r];
    // (a capacity error on the root alone would leave the other ranks inside the collective: the root still takes part, into a scratch buffer)
    DevOwned scratch;
    double *dst = dst_device;
    const bool fits = dst && (size_t)total * 8 <= capacity_bytes;
    if (!fits) { HIPG(hipMalloc(&scratch.p, (size_t)(total ? total : 1) * 8)); dst = static_cast<double *>(scratch.p); }
    {
      RcclGroupGuard grp(R);
      NCCLG(grp.start());
      int64_t off = 0;
      for (int r = 0; r < c->n_ranks; ++r) {
        if (r != root) NCCLG(R.Recv(dst + off, (size_t)cnt[r], ncclDouble, r, c->comm, s->stream));
        off += cnt[r];
      }
      NCCLG(grp.end());
    }
    int64_t off = 0;
    for (int r = 0; r < root; ++r) off += cnt[r];
    HIPG(hipMemcpyAsync(dst + off, s->last_draws, (size_t)mine * 8, hipMemcpyDeviceToDevice, s->stream));
    HIPG(hipStreamSynchronize(s->stream));
    if (!fits) return amwg_fail(AMWG_ESIZE, "amwg_comm_gather_draws: the root's buffer holds %zu bytes, the job's draws are %lld", capacity_bytes, (long long)total * 8);
  } else {
    NCCLG(R.Send(s->last_draws, (size_t)mine, ncclDouble, root, c->comm, s->stream));
    HIPG(hipStreamSynchronize(s->stream));
  }
  return AMWG_OK;
}

// Collective: mean and sd over the recorded draws of ALL ranks (the twin of amwg_group_moments: two all-reduces of PR + 1 and PR doubles)
int amwg_comm_moments(amwg_sampler *s, amwg_comm *c, double *mean, double *sd) {
  if (!s || !c) return amwg_fail(AMWG_EINVAL, "amwg_comm_moments: null argument");
  // (as in amwg_comm_gather_draws: what only this rank can see travels WITH the first all-reduce -- a rank that cannot contribute adds a count of
  // -inf, so that every rank sees a total that is not a count and returns an error, instead of the others waiting in the collective for good)
  char why[200] = "";
  if (!mean || !sd) snprintf(why, sizeof why, "null output");
  else if (!s->last_draws || s->last_rows < 1) snprintf(why, sizeof why, "no sample() call yet");
  else if (s->device != c->device) snprintf(why, sizeof why, "the sampler is on device %d, the communicator on %d", s->device, c->device);
  else if (amwg_sync(s) != AMWG_OK) snprintf(why, sizeof why, "%.190s", amwg_last_error());
  const bool bad = why[0] != 0;
  Rccl &R = rccl();
  const int PR = s->P + s->D;      // (equal on every rank: one model)
  HIPG(hipSetDevice(c->device));
  hipStream_t st = (s->device == c->device) ? s->stream : nullptr;
  DevOwned buf, center;
  HIPG(hipMalloc(&buf.p, (size_t)(PR + 1) * 8));
  HIPG(hipMalloc(&center.p, (size_t)PR * 8));
  double *b = static_cast<double *>(buf.p);
  std::vector<double> h((size_t)PR + 1);
  if (bad) HIPG(hipMemsetAsync(b, 0, (size_t)PR * 8, st));
  else {
    hipLaunchKernelGGL(draw_sums_kernel, dim3(PR), dim3(1024), 0, st, s->last_draws, s->last_rows, PR, s->C, (const double *)nullptr, b);
    HIPG(hipGetLastError());
  }
  const double cnt = bad ? -HUGE_VAL : (double)s->last_rows * (double)s->C;
  HIPG(hipMemcpyAsync(b + PR, &cnt, 8, hipMemcpyHostToDevice, st));
  NCCLG(R.AllReduce(b, b, (size_t)PR + 1, ncclDouble, ncclSum, c->comm, st));
  HIPG(hipMemcpyAsync(h.data(), b, (size_t)(PR + 1) * 8, hipMemcpyDeviceToHost, st));
  HIPG(hipStreamSynchronize(st));
  if (bad) return amwg_fail(AMWG_EINVAL, "amwg_comm_moments: %s (every rank was told)", why);
  if (!(h[PR] > 0)) return amwg_fail(AMWG_EINVAL, "amwg_comm_moments: another rank could not contribute (its own call says why)");
  const double N = h[PR];
  for (int p = 0; p < PR; ++p) mean[p] = h[p] / N;
  HIPG(hipMemcpyAsync(center.p, mean, (size_t)PR * 8, hipMemcpyHostToDevice, s->stream));
  hipLaunchKernelGGL(draw_sums_kernel, dim3(PR), dim3(1024), 0, s->stream, s->last_draws, s->last_rows, PR, s->C, (const double *)center.p, b);
  HIPG(hipGetLastError());
  NCCLG(R.AllReduce(b, b, (size_t)PR, ncclDouble, ncclSum, c->comm, s->stream));
  HIPG(hipMemcpyAsync(h.data(), b, (size_t)PR * 8, hipMemcpyDeviceToHost, s->stream));
  HIPG(hipStreamSynchronize(s->stream));
  for (int p = 0; p < PR; ++p) sd[p] = N > 1 ? std::sqrt(h[p] / (N - 1)) : 0.0;
  return AMWG_OK;
}

}  // extern "C"
