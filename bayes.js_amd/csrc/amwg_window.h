// amwg_window.h -- the chain's uniform stream as a window of 256 uniforms shared by the 64 lanes of a wavefront: what lets the updates of a whole
// sweep draw their proposals side by side (the group-local kernel, amwg_gl.h; the sweep prefetch of the hierarchical family's row layout,
// amwg_kernel.h kSweep).  Same stream as CoopStream / ChainStream (amwg_philox.h).
#pragma once
#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#endif

#include "amwg_math.h"
#include "amwg_philox.h"

namespace amwg {

// ---- the chain's uniform stream as a WINDOW of 256 uniforms: half A = blocks b0 .. b0 + 63 (lane j holds block b0 + j as the two doubles
// it yields), half B = the next 64 blocks, computed when a sweep begins or A runs out.  Same stream as CoopStream / ChainStream; the
// persisted state is still just the number of uniforms consumed.  Both halves are mirrored in LDS (win[0..127] = A, win[128..255] = B) for
// the per-lane reads of a sweep.  E? / O?: bit j = rnorm (mcmc.js:44-53) accepts the pair that starts at the even / odd position 2j / 2j + 1
// of that half; the last odd pair of a half ends in the next one: OA bit 63 is valid once B is, OB bit 63 is never set.
struct WindowStream {
  static constexpr int kLanesPerChain = 64;
  uint32_t k0, k1, c2, c3;
  uint64_t b0;
  uint32_t pos;              // uniforms consumed since block b0 (wave-uniform)
  double a0, a1, b0v, b1v;
  uint64_t EA, OA, EB, OB;
  bool b_valid;
  double *win;               // LDS
  int lane;
  static __device__ __attribute__((noinline)) Philox4 block(uint64_t b, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    return philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), c2, c3, k0, k1);
  }
  // does rnorm accept the pair (u, v)?  mcmc.js:44-53
  static __device__ __forceinline__ bool pair_ok(double u, double v_raw) {
#if defined(AMWG_X_GLCUT) && AMWG_X_GLCUT == 7
    return u > 0.1;
#endif
    const double v = 1.7156 * (v_raw - 0.5);
    const double x = u - 0.449871;
    const double y = __builtin_fabs(v) + 0.386595;
    const double q = x * x + y * (0.19600 * y - 0.25472 * x);
    return !(q > 0.27597 && (q > 0.27846 || v * v > -4 * log_v8_cold(u) * u * u));
  }
  static __device__ __forceinline__ double from_lane(double v, int src) {      // v of lane `src` (wave-uniform), every lane
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(f64_bits(v) >> 32), src);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)f64_bits(v), src);
    return bits_f64(((uint64_t)hi << 32) | (uint64_t)lo);
#else
    return v;
#endif
  }
  static __device__ __forceinline__ double next_lane(double v) {                // v of lane + 1 (lane 63: unspecified), all lanes executing
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __shfl_down(v, 1, 64);
    asm volatile("" : "+v"(r));       // (keeps the shuffle out of a later conditional expression: a masked-off source lane reads as 0)
    return r;
#else
    return v;
#endif
  }
  __device__ __forceinline__ void store_a() { win[2 * lane] = a0; win[2 * lane + 1] = a1; }
  __device__ __forceinline__ void store_b() { win[128 + 2 * lane] = b0v; win[128 + 2 * lane + 1] = b1v; }
  __device__ __forceinline__ void init(uint64_t seed, uint64_t chain, uint64_t consumed, int tid, double *window) {
    k0 = (uint32_t)seed; k1 = (uint32_t)(seed >> 32);
    c2 = (uint32_t)chain; c3 = (uint32_t)(chain >> 32);
    lane = tid & 63;
    win = window;
    b0 = consumed >> 1;
    pos = (uint32_t)(consumed & 1u);
    const Philox4 w = block(b0 + (uint64_t)lane, c2, c3, k0, k1);
    a0 = u53(w.w0, w.w1); a1 = u53(w.w2, w.w3);
    store_a();
    EA = __ballot(pair_ok(a0, a1));
    const double na = next_lane(a0);
    OA = __ballot(pair_ok(a1, na)) & ~(1ull << 63);
    EB = OB = 0ull;
    b0v = b1v = 0.0;
    b_valid = false;
  }
  // half B: one Philox block per lane, its flags, and the one pair of A that ends in it
  __device__ __forceinline__ void ensure_b() {
    if (b_valid) return;
    const Philox4 w = block(b0 + 64ull + (uint64_t)lane, c2, c3, k0, k1);
    b0v = u53(w.w0, w.w1); b1v = u53(w.w2, w.w3);
    store_b();
    EB = __ballot(pair_ok(b0v, b1v));
    // odd pairs: lanes 0..62 their own second uniform with the next lane's first; lane 63 does A's last odd pair with B's first uniform
    const double nb = next_lane(b0v);
    const double b_first = from_lane(b0v, 0);
    const double pu = lane == 63 ? a1 : b1v, pv = lane == 63 ? b_first : nb;
    const uint64_t m = __ballot(pair_ok(pu, pv));
    OB = m & ~(1ull << 63);
    OA = (OA & ~(1ull << 63)) | (m & (1ull << 63));
    b_valid = true;
  }
  // A is used up: B becomes A
  __device__ __forceinline__ void shift() {
    ensure_b();
    b0 += 64ull;
    a0 = b0v; a1 = b1v;
    EA = EB; OA = OB;
    store_a();
    b_valid = false;
  }
  __device__ __forceinline__ uint64_t consumed() const { return 2 * b0 + (uint64_t)pos; }
  __device__ __forceinline__ uint32_t position() {      // < 128 afterwards: at least 128 uniforms of the window lie ahead once B is there
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t p = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
#else
    uint32_t p = pos;
#endif
    while (p >= 128u) { shift(); p -= 128u; }
    pos = p;
    return p;
  }
  __device__ __forceinline__ double next() {
    const uint32_t p = position();
    const double mine = (p & 1u) ? a1 : a0;
    pos = p + 1u;
    return from_lane(mine, (int)(p >> 1));
  }
  // uniform number q of the window (0 .. 255), per lane
  __device__ __forceinline__ double at(uint32_t q) const { return win[q & 255u]; }
};

}  // namespace amwg
