// amwg_philox.h -- per-chain counter-based RNG (Philox4x32-10, Salmon et al. SC'11).
//
// Replaces every Math.random() on the path (mcmc.js:46, 47, 230, 528).  One sequential
// stream of uniforms per chain:
//     block b of global chain c under seed s = philox4x32_10(ctr = {b_lo, b_hi, c_lo, c_hi}, key = {s_lo, s_hi})
//     uniform #2b = top 53 bits of (r0:r1) * 2^-53,   uniform #2b+1 = top 53 bits of (r2:r3) * 2^-53
// The JS twin used to seed the reference (oracle/philox.js) and the C oracle follow the
// same contract, which is what makes accept decisions comparable one by one.
#pragma once
#include "amwg_math.h"

namespace amwg {

struct Philox4 { uint32_t w0, w1, w2, w3; };

AMWG_HD uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

AMWG_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32x32->64 product per multiplier (v_mad_u64_u32 gives both halves; integer multiplies are quarter rate)
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    c0 = hi1 ^ c1 ^ k0;
    c2 = hi0 ^ c3 ^ k1;
    c1 = lo1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}

AMWG_HD double u53(uint32_t hi, uint32_t lo) {
  return (double)(((uint64_t)hi << 21) | (uint64_t)(lo >> 11)) * 1.1102230246251565e-16;  // * 2^-53, exact
}

// Sequential view of one chain's stream.  `n` = uniforms consumed so far (persisted per chain).
struct ChainStream {
  uint32_t k0, k1, c2, c3;
  uint64_t n;
  uint32_t r2, r3;  // second half of the current block (valid while n is odd)

  AMWG_HD void init(uint64_t seed, uint64_t chain, uint64_t consumed) {
    k0 = (uint32_t)seed; k1 = (uint32_t)(seed >> 32);
    c2 = (uint32_t)chain; c3 = (uint32_t)(chain >> 32);
    n = consumed;
    r2 = r3 = 0;
    if (n & 1) {  // resuming in the middle of a block
      const uint64_t b = n >> 1;
      const Philox4 w = philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), c2, c3, k0, k1);
      r2 = w.w2; r3 = w.w3;
    }
  }
  AMWG_HD double next() {
    double u;
    if ((n & 1) == 0) {
      const uint64_t b = n >> 1;
      const Philox4 w = philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), c2, c3, k0, k1);
      r2 = w.w2; r3 = w.w3;
      u = u53(w.w0, w.w1);
    } else {
      u = u53(r2, r3);
    }
    ++n;
    return u;
  }
};

}  // namespace amwg
