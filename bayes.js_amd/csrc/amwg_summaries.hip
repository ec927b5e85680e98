// amwg_summaries.hip -- posterior quantiles on the device (part of libamwg.so; separate translation unit because
// the rocPRIM/hipCUB sort templates are slow to compile).  The reference's users compute these in R on the returned
// arrays (tests/test_mcmc_js.R); with 10^5 chains the draws are better summarised where they are.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cmath>

#include "amwg_sampler.h"

namespace {

// draws [row][PR][C] -> contiguous values of one recorded component
__global__ void gather_component_kernel(const double *draws, int64_t rows, int PR, int64_t C, int p, double *out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C) return;
  out[i] = draws[((i / C) * PR + p) * C + (i % C)];
}

// R's default quantile (type 7): h = (n-1) q; x[floor h] + (h - floor h) (x[floor h + 1] - x[floor h])
__global__ void pick_quantiles_kernel(const double *sorted, int64_t n, const double *probs, int n_probs, double *out) {
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

#define HIP_TRYQ(expr)                                                                                   \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) { cleanup(); return amwg_fail(AMWG_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } \
  } while (0)

extern "C" int amwg_last_sample_quantiles(amwg_sampler *s, const double *probs, int32_t n_probs, double *out) {
  if (!s || !probs || !out || n_probs < 1) return amwg_fail(AMWG_EINVAL, "amwg_last_sample_quantiles: bad argument");
  if (!s->last_draws || s->last_rows < 1) return amwg_fail(AMWG_EINVAL, "amwg_last_sample_quantiles: no sample() call yet");
  const int PR = s->P + s->D;
  const int64_t n = s->last_rows * s->C;
  double *vals = nullptr, *sorted = nullptr, *dprobs = nullptr, *dout = nullptr;
  void *tmp = nullptr;
  auto cleanup = [&]() { (void)hipFree(vals); (void)hipFree(sorted); (void)hipFree(dprobs); (void)hipFree(dout); (void)hipFree(tmp); };
  HIP_TRYQ(hipSetDevice(s->device));
  HIP_TRYQ(hipMalloc(reinterpret_cast<void **>(&vals), (size_t)n * 8));
  HIP_TRYQ(hipMalloc(reinterpret_cast<void **>(&sorted), (size_t)n * 8));
  HIP_TRYQ(hipMalloc(reinterpret_cast<void **>(&dprobs), (size_t)n_probs * 8));
  HIP_TRYQ(hipMalloc(reinterpret_cast<void **>(&dout), (size_t)n_probs * 8));
  HIP_TRYQ(hipMemcpyAsync(dprobs, probs, (size_t)n_probs * 8, hipMemcpyHostToDevice, s->stream));
  size_t tmp_bytes = 0;
  HIP_TRYQ(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, vals, sorted, (int)n, 0, 64, s->stream));
  HIP_TRYQ(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 8));
  if (n > 2147483647) { cleanup(); return amwg_fail(AMWG_EINVAL, "amwg_last_sample_quantiles: more than 2^31 values per component"); }
  for (int p = 0; p < PR; ++p) {
    hipLaunchKernelGGL(gather_component_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, s->last_draws, s->last_rows, PR, s->C, p, vals);
    HIP_TRYQ(hipGetLastError());
    HIP_TRYQ(hipcub::DeviceRadixSort::SortKeys(tmp, tmp_bytes, vals, sorted, (int)n, 0, 64, s->stream));
    hipLaunchKernelGGL(pick_quantiles_kernel, dim3((unsigned)((n_probs + 63) / 64)), dim3(64), 0, s->stream, sorted, n, dprobs, n_probs, dout);
    HIP_TRYQ(hipGetLastError());
    HIP_TRYQ(hipMemcpyAsync(out + (size_t)p * n_probs, dout, (size_t)n_probs * 8, hipMemcpyDeviceToHost, s->stream));
  }
  HIP_TRYQ(hipStreamSynchronize(s->stream));
  cleanup();
  return AMWG_OK;
}
