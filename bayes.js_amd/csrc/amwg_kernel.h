// amwg_kernel.h -- the fused many-chain AMWG step kernel for gfx950.
//
// One launch advances every chain by n_steps Sampler.step()s (mcmc.js:985-997).  Fused per
// chain: the persistent shuffle of the named sub-steppers (mcmc.js:886-892), the per-step
// shuffle of a multidimensional parameter's components (mcmc.js:685-688, 244-263), and for
// every scalar component the whole OnedimMetropolisStepper.step (mcmc.js:517-553): Philox ->
// rnorm proposal (mcmc.js:43-54, 577-579, 596-598) -> bounds -> log_post of the proposal (the
// user's closure + ld.*) -> exp(delta) > u -> Roberts-Rosenthal batch adaptation of the
// log proposal SD, plus the optional draw write-back of Sampler.sample (mcmc.js:1020-1027).
//
// Mapping (CDNA4, wave64): a chain is owned by G consecutive lanes of one wavefront,
// G in {1,2,4,...,64}.  All G lanes run the chain's scalar logic redundantly (same stream,
// same decisions -- no broadcast needed); the observation loop of log_post is split G ways
// (lane j takes observations j, j+G, ...) and the G partial sums are combined with an xor
// butterfly (offsets 1,2,4,..).  G = 64 is "one wavefront per chain"; G = 1 is "one lane
// per chain", whose summation order is exactly the reference's sequential `lp += term`.
// The host picks G so the launch fills the 1024 SIMDs of the chip (amwg_core.hip).
//
// The chain-shared data vector is staged ONCE per launch into LDS with coalesced loads
// (80 KB at cfg2 -- one workgroup of up to 16 waves per CU shares it) and is then read
// conflict-free: the G lanes of a chain read G consecutive elements, the 64/G chains of a
// wave read the same addresses (LDS broadcast).  Per-chain scalar state lives in LDS
// ([chain-in-block][component], odd stride) because the component a lane updates is data dependent.
// Data too large for LDS (cfg5's 3.6 MB design matrix) is read through L2/MALL.
//
// Reference work avoided without changing any result: the reference evaluates log_post
// twice per update (mcmc.js:524-526); the current state's value is cached per chain, which
// is exact because log_post is a pure function of the state.
//
// Since round 3: the kernel is instantiated per workgroup size CLASS (amwg_step_kernel<Model, G, BT>: the register budget of the
// launch it is used for, amwg_kernels.hip); the stepper fetches the next slot's data under the current evaluation, keeps a register
// mirror of the state for the hand-written families, moves its cross-lane traffic (uniform draws, shuffle order, butterfly) off the
// LDS crossbar for a chain on a whole wave, and leaves its cold paths out of line (DESIGN.md section 3 has the list and what each was
// worth); the hierarchical family has an opt-in group-local evaluation with a lane-parallel sweep over theta (second step loop below).
#pragma once
#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#endif

#include "amwg_div.h"
#include "amwg_ld.h"
#include "amwg_philox.h"
#include "amwg_types.h"

namespace amwg {

struct LdsLayout {
  uint32_t data, state, pls, cnt, tot, cc, adapt, pl, idx, perm, total, stride, logpls, bc, xw;
};
// CPB = per-chain state copies in the workgroup: chains per workgroup (lanes per chain <= 64), or -- when one chain spans
// several wavefronts -- one private replica per wavefront (`multi`, see step_body).
__host__ __device__ inline LdsLayout lds_layout(size_t data_bytes, int P, int CPB, int max_top, int n_named, bool multi = false) {
  LdsLayout L;
  uint32_t o = 0;
  L.stride = (uint32_t)P | 1u;  // doubles per chain, odd (see StateView)
  L.data = o;  o += (uint32_t)((data_bytes + 15) & ~(size_t)15);
  L.state = o; o += L.stride * CPB * 8;
  L.pls = o;   o += L.stride * CPB * 8;   // proposal sd = exp(prop_log_scale), same [chain][stride] layout as the state
  L.cnt = o;   o += L.stride * CPB * 8;   // {acceptance_count, iterations_since_adaption} int32 pairs
  L.tot = o;   o += (L.stride * CPB * 4 + 15) & ~15u;   // this launch's run totals, packed: accepts << 16 | evaluated (in-bounds) proposals
  L.cc = o;    o += (uint32_t)P * sizeof(CompConst);
  L.adapt = o; o += ((uint32_t)P + 7) & ~7u;
  L.pl = o;    o += (uint32_t)n_named * 20u;            // base | len | top | multidim | inner (= len / top), int32 each
  o = (o + 15) & ~15u;
  L.idx = o;   o += (max_top > 1) ? (((uint32_t)max_top * CPB * (max_top > kByteTop ? 2u : 1u) + 15) & ~15u) : 0;   // shuffle indices, u8 or u16
  L.perm = o;  o += (n_named > kPackedNamed) ? (((uint32_t)n_named * CPB * 2u + 15) & ~15u) : 0;                  // order of the named steppers, u16
  L.logpls = o; o += multi ? L.stride * CPB * 8 : 0;   // prop_log_scale replicas (single-wave chains keep it in HBM)
  L.bc = o;     o += multi ? ((L.stride * CPB * 4 + 15) & ~15u) : 0;   // batch_count replicas
  L.xw = o;     o += multi ? 2u * 16u * 8u : 0;        // cross-wave partial sums, double buffered, <= 16 waves
  L.total = (o + 15) & ~15u;
  return L;
}

template <class Model, bool FAST, int G, int U = Model::kUnroll>
__device__ __forceinline__ double pass_over_data(const typename Model::Pass &ps, int n_obs, int sub, double acc) {
  const int n_full = n_obs / G, rem = n_obs % G;
  int k = 0;
  for (; k + U <= n_full; k += U) {
    double t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) t[u] = Model::template term<FAST>(ps, (k + u) * G + sub);
#pragma unroll
    for (int u = 0; u < U; ++u) acc += t[u];
  }
  for (; k < n_full; ++k) acc += Model::template term<FAST>(ps, k * G + sub);
  if (sub < rem) acc += Model::template term<FAST>(ps, n_full * G + sub);
  return acc;
}

// log_post(state) in the documented order: lane 0 of the chain starts from the prior sum
// (accumulated sequentially as the closure does), every lane adds its observations in
// increasing index order, then the xor butterfly.  For G = 1 this is the reference's order.
// Exchange area of a chain that spans several wavefronts (lanes per chain G > 64): every wave leaves its partial sum
// in LDS, ONE barrier, then every wave adds the partials in the same xor-butterfly order (offsets 64, 128, ...), so all
// G lanes end up with the same bits.  The buffer alternates between two halves: a wave that races ahead into the next
// evaluation writes the other half, and cannot come back to this one before everybody passed the next barrier.
struct CrossWave {
  double *buf;     // [2][16]
  int parity;
};

template <class...> using void_of = void;

// Does the model have binary parameters (BinaryStepper, two more inlined log_post evaluations per slot)?  The built-in families have
// none; a translated closure says so (UserModel::kHasBinary, from its params); absent means "may have".
template <class M, class = void> struct BinaryOf { static constexpr bool value = true; };
template <class M> struct BinaryOf<M, void_of<decltype(M::kHasBinary)>> { static constexpr bool value = M::kHasBinary; };

// Does the model mirror the chain's state in registers (Model::kTracksState; translated closures read the LDS copy)?  Such a model is
// told about every store to the state -- on_set(cache, component, value, lane, data) -- so that its log_post never has to wait for an
// LDS round trip to learn what the stepper has just written.
template <class M, class = void> struct TracksState { static constexpr bool value = false; };
template <class M> struct TracksState<M, void_of<decltype(M::kTracksState)>> { static constexpr bool value = M::kTracksState; };

// Does the model keep per-lane sums across evaluations (Model::kLaneReuse; HierNormalModel's row layout)?
template <class M, class = void> struct LaneReuseOf { static constexpr bool value = false; };
template <class M> struct LaneReuseOf<M, void_of<decltype(M::kLaneReuse)>> { static constexpr bool value = M::kLaneReuse; };

// Does the model offer a cheaper value of log_post with a bound on its distance from the expression's (Model::kCertified, log_post_approx)?
template <class M, class = void> struct CertifiedOf { static constexpr bool value = false; static constexpr int lanes = 0; };
template <class M> struct CertifiedOf<M, void_of<decltype(M::kCertified)>> { static constexpr bool value = M::kCertified; static constexpr int lanes = M::kCertifiedLanes; };
template <class M, int G, class = void> struct CertifiedAt { static constexpr bool value = false; };
template <class M, int G> struct CertifiedAt<M, G, void_of<decltype(M::kCertified)>> { static constexpr bool value = M::kCertified && M::kCertifiedLanes == G; };

// ... only in its row layout, i.e. in the sweep kernel (Model::kCertifiedNeedsRows)?
template <class M, class = void> struct CertNeedsRows { static constexpr bool value = false; };
template <class M> struct CertNeedsRows<M, void_of<decltype(M::kCertifiedNeedsRows)>> { static constexpr bool value = M::kCertifiedNeedsRows; };

// ... and can the model evaluate the expression in the REFERENCE's order (one running sum) at this lane count (Model::kReferenceOrder, reference_order)?  Then the
// certified kernels decide against THAT expression: accept counts identical to the reference's whatever the lane count
template <class M, class = void> struct RefOrderOf { static constexpr bool value = false; };
template <class M> struct RefOrderOf<M, void_of<decltype(M::kReferenceOrder)>> { static constexpr bool value = M::kReferenceOrder; };

template <class M, class = void> struct OwnPassOf { static constexpr bool value = false; };
template <class M> struct OwnPassOf<M, void_of<decltype(M::kOwnPass)>> { static constexpr bool value = M::kOwnPass; };

// per-chain values a model keeps from one log_post evaluation to the next (Model::Cache; translated closures have none)
struct NoCache {};
template <class M, class = void> struct CacheOf { using type = NoCache; static __device__ __forceinline__ type init() { return type{}; } };
template <class M> struct CacheOf<M, void_of<typename M::Cache>> { using type = typename M::Cache; static __device__ __forceinline__ type init() { return M::cache_init(); } };

// The value a lane of the xor butterfly adds at offset OFF: that of lane ^ OFF or, for OFF = 4 and 8, of another lane of the same partner
// group (which holds the same bits at that stage).  Offsets 1, 2 (quad permutes), 4 and 8 (row mirrors) are single DPP moves in the VALU --
// no trip through the LDS crossbar; 16 and 32 are handled in xor_sum.  Every lane adds the same two numbers as with __shfl_xor.
template <int OFF, bool EXACT = false>
__device__ __forceinline__ double xor_partner(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  int lo = (int)(uint32_t)f64_bits(v), hi = (int)(uint32_t)(f64_bits(v) >> 32);
  if constexpr (OFF == 1) {          // quad_perm [1,0,3,2]
    lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  } else if constexpr (OFF == 2) {   // quad_perm [2,3,0,1]
    lo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true);
  } else if constexpr (OFF == 4) {   // row_half_mirror (i -> 7 - i within 8): a lane of the partner quad -- inside the butterfly all four lanes of a
                                     // quad hold the same sum by now (IEEE addition is commutative), so any of them is "the" partner
    lo = __builtin_amdgcn_mov_dpp(lo, 0x141, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x141, 0xF, 0xF, true);
    if constexpr (EXACT) {           // lane ^ 4 itself (a partial butterfly, whose lanes do not hold equal values yet): then quad_perm [3,2,1,0]
      lo = __builtin_amdgcn_mov_dpp(lo, 0x1B, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x1B, 0xF, 0xF, true);
    }
  } else if constexpr (OFF == 8) {   // row_mirror (i -> 15 - i within 16): a lane of the partner group of eight, likewise
    lo = __builtin_amdgcn_mov_dpp(lo, 0x140, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x140, 0xF, 0xF, true);
    if constexpr (EXACT) {           // lane ^ 8 itself: then row_half_mirror
      lo = __builtin_amdgcn_mov_dpp(lo, 0x141, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x141, 0xF, 0xF, true);
    }
  } else if constexpr (OFF == 16) {  // ds_swizzle, bit-mask mode: lane' = (lane & 0x1f) ^ 0x10 within each half of the wave
    lo = __builtin_amdgcn_ds_swizzle(lo, 0x401F); hi = __builtin_amdgcn_ds_swizzle(hi, 0x401F);
  } else {
    return __shfl_xor(v, OFF, 64);
  }
  return bits_f64(((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo);
#else
  return v;
#endif
}
// acc + (the value of lane ^ OFF), every lane.  Offsets 16 and 32 are gfx950's v_permlane16_swap / v_permlane32_swap: with both operands the
// same register the instruction leaves "my half / row" in one result and "the partner half / row" in the other (which is which depends on
// the lane, and does not matter: IEEE addition is commutative), so the sum of the two results is acc + partner in every lane -- two VALU
// moves per 32-bit half and no trip through the LDS crossbar (round 2: ds_swizzle and ds_bpermute, i.e. two dependent LDS round trips per
// evaluation queued behind the data passes of the CU's other waves).
template <int OFF, bool EXACT = false>
__device__ __forceinline__ double xor_sum(double acc) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (OFF == 16 || OFF == 32) {
    const uint32_t lo = (uint32_t)f64_bits(acc), hi = (uint32_t)(f64_bits(acc) >> 32);
    if constexpr (OFF == 32) {
      const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
      return bits_f64(((uint64_t)h[0] << 32) | (uint64_t)l[0]) + bits_f64(((uint64_t)h[1] << 32) | (uint64_t)l[1]);
    } else {
      const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
      return bits_f64(((uint64_t)h[0] << 32) | (uint64_t)l[0]) + bits_f64(((uint64_t)h[1] << 32) | (uint64_t)l[1]);
    }
  } else {
    return acc + xor_partner<OFF, EXACT>(acc);
  }
#else
  return acc;
#endif
}
template <int OFF, int LIMIT>
__device__ __forceinline__ double butterfly(double acc) {
  if constexpr (OFF < LIMIT) { acc = xor_sum<OFF>(acc); return butterfly<OFF * 2, LIMIT>(acc); }
  else return acc;
}

#ifndef AMWG_STEPPER_PRIORITY
#define AMWG_STEPPER_PRIORITY 2
#endif
constexpr int kStepperPriority = AMWG_STEPPER_PRIORITY;
__device__ __forceinline__ void wave_priority(int p) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (kStepperPriority > 0) { if (p == 0) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(kStepperPriority); }
#endif
}

// a wave-uniform value the compiler must treat as freshly defined HERE: everything derived from it (loop bounds, block counts, tail masks,
// base addresses) is then worked out inside the evaluation on the scalar unit -- a few SALU instructions per evaluation, issued beside the
// other wave's vector work -- instead of being hoisted out of the step loop and carried in scalar registers across all of it
__device__ __forceinline__ int fresh_uniform(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
#if !defined(AMWG_X_NOFRESH)
  v = __builtin_amdgcn_readfirstlane(v);      // (folds away when the value is already in a scalar register)
  asm volatile("" : "+s"(v));
#endif
#endif
  return v;
}

template <class Model, int G, int U = 8>
__device__ __forceinline__ double log_post(const StateView &S, const StepArgs &a0, const unsigned char *smem, int sub, CrossWave &xw,
                                           typename CacheOf<Model>::type &cache) {
  // (a copy of the argument block's data descriptor with opaque sizes, see fresh_uniform)
  struct { const ModelConsts &mc; DataRef d; } a{a0.mc, a0.d};
  a.d.n_obs = fresh_uniform(a0.d.n_obs);
  a.d.G = fresh_uniform(a0.d.G);
  double acc;
  if constexpr (Model::kUser) {
    // translated closure: the generated body returns this lane's partial sum (lane 0 carries every
    // term outside the lane-split loops), see bayes.js_amd/translate.js
    wave_priority(0);      // (the whole evaluation of a translated closure counts as "the data pass" for the issue priority, see below)
    bool summed = false;
    if constexpr (LaneReuseOf<Model>::value && G == 64) {
      // a closure that ends in a likelihood loop with group means, in the row layout (amwg_rows.h): the lanes whose three numbers did not change keep their sums
      if (a0.d.pad > 0) { acc = Model::template rows_eval<U>(cache, S, a.d, smem, sub, a0.d.pad, (int)(threadIdx.x >> 6)); summed = true; }
    }
    if (!summed) acc = Model::template eval<G, false>(S, a.d, smem, sub, nullptr);
  } else {
    if constexpr (TracksState<Model>::value) Model::template load<G>(cache, S, a.mc, a.d, smem, sub);   // first evaluation: fill the register mirror
    const typename Model::Pass ps = Model::template begin<G>(S, a.mc, a.d, smem, cache);
    const double prior = Model::prior(S, a.mc, a.d, cache);
    acc = (sub == 0) ? prior : 0.0;
    if constexpr (Model::kSplitPrior) acc = Model::template prior_split<G>(S, a.mc, a.d, sub, acc, cache);
    // Issue priority: the stepper around this point is one long dependent chain (a wave alone issues an instruction every ~8 cycles in it),
    // the data pass below is hundreds of independent instructions.  With equal priorities the SIMD's arbiter favours the OLDER wave, so a
    // younger wave's stepper starves behind an older wave's pass and the two waves of a SIMD end up in their steppers together, leaving the
    // fp64 pipe half idle.  A wave therefore runs its stepper at raised priority (set at the top of the step loop) and drops to 0 for the
    // pass: whoever is in a stepper issues the moment it can, the partner's pass fills every other slot.
    wave_priority(0);
    bool summed = false;
    if constexpr (LaneReuseOf<Model>::value && G == 64) {
      // row layout: the lanes whose sum cannot have changed keep it, the others are re-formed (amwg_models.h lane_sum_rows)
      if (ps.rows) { acc = Model::template lane_sum_rows<U>(cache, ps, acc, a.d.n_obs, a.d.G, sub, smem, a0.d.pad, (int)(threadIdx.x >> 6)); summed = true; }
    }
    if (summed) {
    } else if constexpr (Model::kHasFast) {
      if (ps.fast) acc = Model::template pass_fast<G, U>(ps, a.d.n_obs, sub, acc);   // hand-pipelined (amwg_models.h norm_pass_staged)
      else acc = Model::template pass_slow<G>(ps, a.d.n_obs, sub, acc);            // IEEE '/': rare, out of line
    } else if constexpr (Model::kOneLanePass && G == 1) {
      acc = Model::pass_one_lane(ps, a.d.n_obs, acc);
    } else if constexpr (OwnPassOf<Model>::value) {
      acc = Model::template pass<G>(ps, a.d.n_obs, sub, acc);      // the model's own pipelined pass
    } else {
      acc = pass_over_data<Model, false, G>(ps, a.d.n_obs, sub, acc);
    }
    if constexpr (Model::kOneLanePass) {
      if (ps.has_invalid) acc = acc + (-kInf);   // some x_i outside {0,1}: that term is -inf wherever it sits in the sum
    }
  }
  wave_priority(kStepperPriority);
  acc = butterfly<1, (G < 64 ? G : 64)>(acc);
  if constexpr (G > 64) {
    constexpr int WV = G / 64;
    double *slot = xw.buf + xw.parity * 16;
    xw.parity ^= 1;
    const int wave = sub >> 6;
    if ((sub & 63) == 0) slot[wave] = acc;
    __syncthreads();
    double t[WV];
#pragma unroll
    for (int w = 0; w < WV; ++w) t[w] = slot[w];
#pragma unroll
    for (int off = 1; off < WV; off <<= 1) {
      double u[WV];
#pragma unroll
      for (int w = 0; w < WV; ++w) u[w] = t[w] + t[w ^ off];
#pragma unroll
      for (int w = 0; w < WV; ++w) t[w] = u[w];
    }
    acc = t[0];   // every entry holds the same sum
  }
  return acc;
}

// The chain's uniform stream, shared by the L = min(G, 64) lanes that run the chain inside one wave.  Every lane needs every
// uniform (the scalar logic is replicated), but a Philox block is ~90 instructions: instead of all L lanes computing the SAME
// block, lane j computes block b0 + j, and uniform #n is fetched from the lane that holds block n >> 1 with a cross-lane
// permute -- one Philox evaluation per lane buys 2L uniforms for the chain.  Same stream as ChainStream (amwg_philox.h), the
// persisted state is still just the number of uniforms consumed.
template <int G>
struct CoopStream {
  static constexpr int L = G < 64 ? G : 64;
  static constexpr int kLanesPerChain = G;
  uint32_t k0, k1, c2, c3;
  uint64_t b0;          // first block held by the chain's lanes
  uint32_t pos;         // uniforms consumed since block b0 (0 .. 2L): the stream position is 2*b0 + pos -- 32-bit bookkeeping per draw
  double u0, u1;        // this lane's block as the two uniforms it yields (u53 of the words), converted ONCE per fill: every draw is then a
                        // cross-lane read of a finished double -- the conversion (2 cvt, ldexp, add, ldexp) used to sit on the dependent chain of
                        // every one of the stepper's ~4 draws per update
  int lane_in_chain, base_lane;
  // one Philox block per lane, OUT OF LINE: ~85 instructions that run once per 2L uniforms but would otherwise be inlined at every one of
  // the stepper's eight draw sites (a tenth of the kernel's code, all of it on the path the instruction cache has to hold)
  static __device__ __attribute__((noinline)) Philox4 block(uint64_t b, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    return philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), c2, c3, k0, k1);
  }
  __device__ __forceinline__ void set_block(const Philox4 &w) { u0 = u53(w.w0, w.w1); u1 = u53(w.w2, w.w3); }
  __device__ __forceinline__ void fill() { set_block(block(b0 + (uint64_t)lane_in_chain, c2, c3, k0, k1)); }
  __device__ __forceinline__ void init(uint64_t seed, uint64_t chain, uint64_t consumed, int tid) {
    k0 = (uint32_t)seed; k1 = (uint32_t)(seed >> 32);
    c2 = (uint32_t)chain; c3 = (uint32_t)(chain >> 32);
    lane_in_chain = tid & (L - 1);
    base_lane = (tid & 63) & ~(L - 1);
    b0 = consumed >> 1;
    pos = (uint32_t)(consumed & 1u);
    fill();
  }
  __device__ __forceinline__ uint64_t consumed() const { return 2 * b0 + (uint64_t)pos; }
  __device__ __forceinline__ double next() {
    if constexpr (L == 64) {
      // the chain IS the wave: the stream position is wave-uniform, so the lane that holds the block is named by a scalar and its uniform
      // is read with v_readlane -- no trip through the LDS crossbar (round 2: two ds_bpermute per uniform, a dependent LDS round trip
      // queued behind the data passes of the CU's other waves, on the critical path of every proposal)
      uint32_t p = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
      if (p >= 128u) { b0 += 64ull; p = 0u; fill(); }
      const double mine = (p & 1u) ? u1 : u0;
      const int src = (int)(p >> 1);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(f64_bits(mine) >> 32), src);
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)f64_bits(mine), src);
      pos = p + 1u;
      return bits_f64(((uint64_t)hi << 32) | (uint64_t)lo);
    } else {
      if (pos >= 2u * (uint32_t)L) { b0 += (uint64_t)L; pos = 0u; fill(); }      // pos only ever reaches 2L exactly
      double mine = (pos & 1u) ? u1 : u0;
      if constexpr (L > 1) mine = __shfl(mine, base_lane + (int)(pos >> 1), 64);
      ++pos;
      return mine;
    }
  }
};

// A condition every lane of the chain decides alike.  For a chain on a whole wave (G >= 64) saying so -- through the ballot, which is a scalar -- turns
// the branch into a scalar one: no exec-mask save / restore, no per-lane merges of the values defined under it.  (The compiler cannot know that the
// 64 lanes hold the same chain: every `if` on a value computed in vector registers otherwise compiles to the divergent form.)
template <int G>
__device__ __forceinline__ bool chain_true(bool c) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (G >= 64) return __ballot(c) != 0ull;
#endif
  return c;
}

// THE CERTIFIED TEST (both sites of step_body's certified decisions).  The difference D of the expression's two values lies within eta of dA, the difference of
// the cheap ones (eta: the two bounds + the subtraction's rounding, x 1.0625, + 2^-49).  Is exp_v8(D) > u (mcmc.js:527-528)?  +1: yes for every such D; -1: no
// for every such D; 0: cannot tell -- the expression decides.
//   eta < 2^-7 (all but pathological states): exp(D) lies within a factor 1 -+ eta of exp(dA) (exp(e) <= 1 + 1.0625 e there; the 2^-49 covers V8's exp being within
//   an ulp of exp): one exponential, `ex`, which the caller has.  u = 0 (one uniform in 2^53) asks "is the exponential positive": answered only far from
//   the underflow threshold.
//   Wider bounds (out of line): the hopeless and the certain first -- dA + eta < -746: the exponential is exactly 0 (the stepper's own shortcut); dA - eta >= 0: it
//   is >= 1 > u.  That is what a proposal far out in the tails needs: a sigma a thousand times too small has |log_post| ~ 1e10 and a bound to match, and is rejected
//   all the same -- until this was added such proposals (1e-4 of sigma's while the proposal scales are not yet adapted) each cost an evaluation of the expression in
//   the reference's order, and a launch waits for its slowest wavefront: 1.3 ms instead of 0.5 per 25 steps of cfg4.  In between: exp_v8(dA -+ eta) against u with
//   2^-50 of slack for the two exponentials' own ulps.
__device__ inline __attribute__((noinline)) int certified_test_wide(double dA, double eta, double u) {
  const double lo = dA - eta, hi = dA + eta;
  if (hi < -746.0) return -1;
  if (lo >= 0.0) return 1;
  if (!(eta < 0x1p+9) || !(u > 0x1p-60)) return 0;      // (a NaN; a bound wider than the exponential's whole range; u == 0)
  if (exp_v8(lo) * (1.0 - 0x1p-50) > u) return 1;
  if (exp_v8(hi) * (1.0 + 0x1p-50) < u) return -1;
  return 0;
}
__device__ __forceinline__ int certified_test(double dA, double eta, double ex, double u) {
  if (eta < 0x1p-7) return (ex * (1.0 - eta) > u && (u > 0.0 || dA > -700.0)) ? 1 : (ex * (1.0 + eta) < u ? -1 : 0);
  return certified_test_wide(dA, eta, u);
}

template <class Rng>
__device__ __forceinline__ double rnorm_js(Rng &rng, double mean, double sd) {  // mcmc.js:43-54
  constexpr int GU = Rng::kLanesPerChain;      // (a chain on a whole wave: the loop and its inner test are scalar branches, see chain_true)
  double u, v, q;
  bool again;
  do {
    u = rng.next();
    v = 1.7156 * (rng.next() - 0.5);
    const double x = u - 0.449871;
    const double y = __builtin_fabs(v) + 0.386595;
    q = x * x + y * (0.19600 * y - 0.25472 * x);
    again = false;
    if (chain_true<GU>(q > 0.27597)) again = chain_true<GU>(q > 0.27846) || chain_true<GU>(v * v > -4 * log_v8_cold(u) * u * u);
  } while (again);
  return (v / u) * sd + mean;
}

// Shuffle index tables in LDS, one column per chain of the workgroup (entry t of this chain at [t * CPB]); bytes while every index
// fits one, 16-bit entries otherwise (a leading dimension > 256, more than 16 named parameters).  `wide` is uniform over the launch.
struct IndexColumn {
  unsigned char *base;    // already offset to this chain's column
  int CPB;
  bool wide;
  __device__ __forceinline__ int get(int t) const {
    return wide ? (int)reinterpret_cast<const uint16_t *>(base)[t * CPB] : (int)base[t * CPB];
  }
  __device__ __forceinline__ void set(int t, int v) const {
    if (wide) reinterpret_cast<uint16_t *>(base)[t * CPB] = (uint16_t)v; else base[t * CPB] = (uint8_t)v;
  }
};
__device__ __forceinline__ uint32_t perm_get(uint64_t perm, int i) { return (uint32_t)(perm >> (4 * i)) & 0xFu; }
__device__ __forceinline__ uint64_t perm_swap(uint64_t perm, int i, int j) {
  const uint64_t d = ((perm >> (4 * i)) ^ (perm >> (4 * j))) & 0xFull;
  return perm ^ (d << (4 * i)) ^ (d << (4 * j));
}

// Math.max of two numbers (mcmc.js:758): NaN if either is NaN, +0 > -0.
__device__ __forceinline__ double js_max2(double a, double b) {
  if (a != a || b != b) return __builtin_nan("");
  if (a == b) return (a == 0 && __builtin_signbit(a)) ? b : a;
  return a > b ? a : b;
}

// the random stream a step kernel draws from: the cooperative stream of the chain's lanes, or -- group-local kernel -- the model's window stream
template <class M, int G, bool GL, bool SW = false> struct RngOf { using type = CoopStream<G>; };
template <class M, int G> struct RngOf<M, G, true, false> { using type = typename M::Stream; };
template <class M, int G> struct RngOf<M, G, false, true> { using type = typename M::SweepStream; };      // the sweep kernel (kSweep): the window stream as well
// bytes of the data region of a workgroup's LDS
template <class M, class = void> struct DynamicLdsOf { static constexpr bool value = false; };
template <class M> struct DynamicLdsOf<M, void_of<decltype(M::kDynamicLds)>> { static constexpr bool value = M::kDynamicLds; };
template <class M, bool GL> struct DataBytesOf {
  static __device__ __forceinline__ size_t get(const DataRef &d, int lanes, int threads) {
    if constexpr (DynamicLdsOf<M>::value) return M::lds_bytes_of(d, lanes, threads);      // (a layout the host chose for this launch geometry, DataRef::pad)
    else return M::lds_bytes(d.n_obs, d.G, lanes);
  }
};
template <class M> struct DataBytesOf<M, true> { static __device__ __forceinline__ size_t get(const DataRef &d, int, int threads) { return M::gl_lds_bytes(d.pad, threads / 64); } };

// The kernel's argument block, read on demand.  Arguments taken by value are all loaded into scalar registers in the kernel's entry
// block and stay there until their last use: the dozen per-chain array pointers needed again only by the write-back after the step loop
// (and at batch boundaries / recorded steps inside it) alone are ~30 SGPRs held across the whole loop, and round 2's kernels carried ~200
// spilled SGPRs through v_writelane / v_readlane.  cold_args() hands out the same block through the kernarg segment pointer, made opaque
// at the point of use, so that those fields are fetched (s_load, scalar cache) where they are needed and live nowhere else.
// Precondition: StepArgs is the kernel's ONLY parameter (true for amwg_step_kernel and for the hiprtc-compiled amwg_user_step).
#if defined(__HIP_DEVICE_COMPILE__)
typedef const StepArgs __attribute__((address_space(4))) *cold_args_ptr;
__device__ __forceinline__ cold_args_ptr cold_args() {
  cold_args_ptr p = (cold_args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}
#else
typedef const StepArgs *cold_args_ptr;
__device__ inline cold_args_ptr cold_args() { return nullptr; }
#endif

// What a slot needs to know about its component, fetched from LDS ONE SLOT AHEAD: which component the next slot updates does not
// depend on the current accept decision (within a step every component is visited exactly once), so its constants, current value,
// proposal sd and batch counters are requested before the current evaluation starts and have landed long before they are used --
// round 2 looked them up at the top of the slot, three dependent LDS round trips on the critical path of every update.
struct SlotPre {
  int comp;
  double cur, sd, batch_size;
  int2 cnt;
  bool adapting;
};

__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {      // a wave-uniform 64-bit value, in scalar registers
#if defined(__HIP_DEVICE_COMPILE__)
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
#else
  return v;
#endif
}
// lane `lane` of v = value (both wave-uniform), the other lanes unchanged: one v_writelane_b32.  The lane select has to travel in M0 (a VOP3
// instruction of gfx9 reads one scalar register; this compiler has no builtin for the instruction), which is saved and restored around it.
__device__ __forceinline__ int write_lane(int v, int value, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
  int keep;
  asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1" : "+v"(v), "=&s"(keep) : "s"(value), "s"(lane));
#endif
  return v;
}

// The group-local sweep's walk over the uniform stream for components WITHOUT bounds (amwg_gl.h; the general loop is in step_body): update t of
// the order takes the first pair that rnorm accepts at or after stream position p -- a find-first-set on the flag word of p's half (bit 7) and
// parity (bit 0); a pair must start at or before 253 (EB arrives with bit 63 cleared, OB never has it) -- leaves that position in lane t of
// ppv and moves on by 3 (pair + accept uniform).  Stops at `top` updates or when the window is used up (then p = 254 | parity: where the search
// resumes in the next window; a p of 256 or more stays).  Scalar instructions throughout; M0 carries the lane select and is restored.
__device__ __forceinline__ void gl_resolve_unbounded(uint64_t EA, uint64_t OA, uint64_t EB, uint64_t OB, uint32_t &p, int &t, int top, int &ppv) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint64_t m, rest;
  uint32_t tmp, keep_m0;
  asm volatile(
      "s_mov_b32 %[km0], m0\n"
      "1:\n\t"                                   // ---- next update
      "s_cmp_ge_i32 %[t], %[top]\n\t"
      "s_cbranch_scc1 9f\n"
      "2:\n\t"                                   // ---- search from p
      "s_cmp_ge_u32 %[p], 0x100\n\t"
      "s_cbranch_scc1 9f\n\t"
      "s_bitcmp1_b32 %[p], 7\n\t"               // second half?
      "s_cbranch_scc1 3f\n\t"
      "s_bitcmp1_b32 %[p], 0\n\t"
      "s_cselect_b64 %[m], %[OA], %[EA]\n\t"
      "s_branch 4f\n"
      "3:\n\t"
      "s_bitcmp1_b32 %[p], 0\n\t"
      "s_cselect_b64 %[m], %[OB], %[EB]\n"
      "4:\n\t"
      "s_bfe_u32 %[tmp], %[p], 0x60001\n\t"     // (p & 127) >> 1
      "s_lshr_b64 %[rest], %[m], %[tmp]\n\t"
      "s_cmp_lg_u64 %[rest], 0\n\t"
      "s_cbranch_scc1 6f\n\t"
      "s_and_b32 %[tmp], %[p], 1\n\t"           // nothing left in this half
      "s_bitcmp1_b32 %[p], 7\n\t"
      "s_cbranch_scc1 5f\n\t"
      "s_or_b32 %[p], %[tmp], 0x80\n\t"         // on to the second half
      "s_branch 2b\n"
      "5:\n\t"
      "s_or_b32 %[p], %[tmp], 0xfe\n\t"         // window used up
      "s_branch 9f\n"
      "6:\n\t"                                   // ---- found
      "s_ff1_i32_b64 %[tmp], %[rest]\n\t"
      "s_lshl_b32 %[tmp], %[tmp], 1\n\t"
      "s_add_u32 %[p], %[p], %[tmp]\n\t"
      "s_mov_b32 m0, %[t]\n\t"
      "v_writelane_b32 %[ppv], %[p], m0\n\t"
      "s_add_u32 %[p], %[p], 3\n\t"
      "s_add_u32 %[t], %[t], 1\n\t"
      "s_branch 1b\n"
      "9:\n\t"
      "s_mov_b32 m0, %[km0]"
      : [p] "+s"(p), [t] "+s"(t), [ppv] "+v"(ppv), [m] "=&s"(m), [rest] "=&s"(rest), [tmp] "=&s"(tmp), [km0] "=&s"(keep_m0)
      : [EA] "s"(EA), [OA] "s"(OA), [EB] "s"(EB), [OB] "s"(OB), [top] "s"(top)
      : "scc");
#endif
}

// a value every lane of the chain holds alike; for a chain on a whole wave (G >= 64) it is moved to a scalar register, so that what is
// derived from it (table addresses, loop counters, branch conditions) runs on the scalar unit beside the vector work
template <int G>
__device__ __forceinline__ int chain_uniform(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (G >= 64) return __builtin_amdgcn_readfirstlane(v);
#endif
  return v;
}

// the double lane `src` (wave-uniform) holds in v
__device__ __forceinline__ double lane_value(double v, int src) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(f64_bits(v) >> 32), src);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)f64_bits(v), src);
  return bits_f64(((uint64_t)hi << 32) | (uint64_t)lo);
#else
  return v;
#endif
}

// Device-side invariants.  The host keeps them (choose_geometry pairs cpb with one-wavefront workgroups, launch_steps cuts calls into launches
// of at most 65 535 steps); a launch that breaks one -- a future geometry or launch change -- must not look like a successful no-op: the
// kernel leaves a bit in ChainArrays::error, which the host reads after every call (finish_timing) and reports as AMWG_EHIP.
constexpr int kErrReplicasNeedOneWave = 1, kErrLaunchTooLong = 2, kErrMirrorOutOfSync = 4, kErrSweepNeedsOrderInRegisters = 8;
__device__ __forceinline__ void device_error(const StepArgs &a, int code) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (threadIdx.x == 0) (void)atomicOr(a.ch.error, code);
#endif
}
// a model that mirrors the state in registers (kTracksState) says whether the mirror still equals the LDS copy (Model::mirror_ok)
template <class M, class = void> struct MirrorCheckOf { static constexpr bool value = false; };
template <class M> struct MirrorCheckOf<M, void_of<decltype(M::kMirrorCheck)>> { static constexpr bool value = M::kMirrorCheck; };

// BT: the workgroup size class the caller is compiled for (its register budget, see amwg_step_kernel); 1024-thread workgroups leave 128
// VGPRs per lane, and with four waves per SIMD the staged pass does not need eight observations in flight per lane to keep the pipe busy:
// it runs four-wide there (same operations in the same order, half the registers).
template <class Model, int G, int BT = 256, bool GL = false, bool SW = false, bool CT = false>
__device__ __forceinline__ void step_body(const StepArgs &a, unsigned char *smem) {
  // (round 5: in the sweep kernel the expression's passes are the rare path -- certified decisions, amwg_models.h HierNormalModel::sweep_approx --: two-wide, what
  // counts is the registers they leave to everything else)
  constexpr int kPassU = SW ? (CertNeedsRows<Model>::value ? 2 : 4) : (BT >= 1024 ? 4 : 8);      // (the sweep kernel keeps the window stream and the per-lane values of a sweep alive across its passes: four-wide as well -- eight-wide spilled 14 registers)
  const int tid = threadIdx.x, nt = blockDim.x;
  // G <= 64: nt/G chains per workgroup, each on G lanes of one wave.  G > 64 ("multi"): ONE chain per workgroup on G/64
  // waves; every wave is a full replica of the chain's scalar logic (same Philox stream => same proposals and decisions)
  // with its own copy of the stepper state in LDS, so waves share nothing but the data tile and the partial sums.
  constexpr bool kMulti = G > 64;
  // state copies in this workgroup.  A model with hundreds of components and few lanes per chain may not fit blockDim / G copies in
  // LDS (dim [300] with one lane per chain: 64 x 301 x 28 B): the host then passes a smaller a.cpb, and the lane groups beyond it
  // replicate the workgroup's last chain -- same chain id, hence the same stream, decisions and stores: redundant, never different.
  // (Replicas are only sound in lockstep, i.e. inside ONE wavefront: the host pairs cpb with 64-thread workgroups, and a launch that
  // does not is refused here rather than left to race.)
  if (!kMulti && a.cpb > 0 && nt != 64) { device_error(a, kErrReplicasNeedOneWave); return; }
  if (a.n_steps > 65535) { device_error(a, kErrLaunchTooLong); return; }      // run totals of a launch are 16-bit fields (TOTme); the host chunks launches accordingly
  const int CPB = kMulti ? G / 64 : ((a.cpb > 0 && a.cpb < nt / G) ? a.cpb : nt / G);
  const int c_raw = kMulti ? tid / 64 : tid / G;
  const int c_in = c_raw < CPB ? c_raw : CPB - 1, sub = tid % G;
  const int P = a.pl.P;
  const int n_named = a.pl.n_params;
  const LdsLayout L = lds_layout(DataBytesOf<Model, GL>::get(a.d, G, nt), P, CPB, a.pl.max_top, n_named, kMulti);

  const unsigned char *data_lds = smem + L.data;
  double *Sblk = reinterpret_cast<double *>(smem + L.state);
  CompConst *cc = reinterpret_cast<CompConst *>(smem + L.cc);
  uint8_t *adapt = smem + L.adapt;
  int32_t *pl_base = reinterpret_cast<int32_t *>(smem + L.pl), *pl_len = pl_base + n_named, *pl_top = pl_len + n_named, *pl_multidim = pl_top + n_named, *pl_inner = pl_multidim + n_named;
  const bool wide_idx = a.pl.max_top > kByteTop, wide_perm = n_named > kPackedNamed;
  const IndexColumn idx{smem + L.idx + (wide_idx ? 2 * c_in : c_in), CPB, wide_idx};      // this chain's shuffle indices
  const IndexColumn pcol{smem + L.perm + 2 * c_in, CPB, true};                             // order of the named steppers (wide_perm only)
  double *SDme = reinterpret_cast<double *>(smem + L.pls) + (size_t)c_in * L.stride;   // exp(prop_log_scale), mcmc.js:578
  int2 *CNTme = reinterpret_cast<int2 *>(smem + L.cnt) + (size_t)c_in * L.stride;
  // run totals of THIS launch (accepted << 16 | evaluated): kept in LDS and added to the HBM totals once, when the launch ends
  // (round 1 issued two no-return atomics per update: 17x the bytes of the recorded draws).  The host keeps launches <= 65535 steps.
  uint32_t *TOTme = reinterpret_cast<uint32_t *>(smem + L.tot) + (size_t)c_in * L.stride;
  double *LOGPLSme = reinterpret_cast<double *>(smem + L.logpls) + (size_t)c_in * L.stride;   // multi only
  int32_t *BCme = reinterpret_cast<int32_t *>(smem + L.bc) + (size_t)c_in * L.stride;         // multi only
  CrossWave xw{reinterpret_cast<double *>(smem + L.xw), 0};
  typename CacheOf<Model>::type cache = CacheOf<Model>::init();

  // ---- stage chain-shared data and per-component constants (coalesced, once per launch)
  Model::stage(smem + L.data, a.d, tid, nt, G);
  for (int p = tid; p < P; p += nt) { cc[p] = a.cc[p]; adapt[p] = a.is_adapting[p]; }
  for (int i = tid; i < 4 * n_named; i += nt) pl_base[i] = a.pl.tab[i];
  for (int i = tid; i < n_named; i += nt) pl_inner[i] = a.pl.tab[n_named + i] / a.pl.tab[2 * n_named + i];   // len / top: elements per top-level entry

  const int64_t chain_raw = kMulti ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * CPB + c_in;
  const bool live = chain_raw < a.C;
  const int64_t cl = live ? chain_raw : a.C - 1;  // dead lanes shadow the last chain and never store
  const bool writer = live && sub == 0;
  // the one lane that counts this chain's run totals in LDS (a no-return ds_add: the stepper never waits for the old value); the
  // replicas of the cpb fallback and the other waves of a multi-wave chain do not count (wave 0's totals are the ones stored)
  const bool counter = sub == 0 && c_raw < CPB;
  const int64_t C = a.C;

  double *Sme = Sblk + (size_t)c_in * L.stride;
  // per-chain stepper state lives in LDS for the whole launch: an update never waits on HBM/L2
  for (int p = 0; p < P; ++p) {
    Sme[p] = a.ch.state[p * C + cl];
    SDme[p] = exp_v8(a.ch.prop_log_scale[p * C + cl]);   // the log scale itself stays in HBM: it only changes at batch boundaries
    CNTme[p] = make_int2(a.ch.acceptance_count[p * C + cl], a.ch.iterations_since_adaption[p * C + cl]);
    TOTme[p] = 0u;
    if constexpr (kMulti) { LOGPLSme[p] = a.ch.prop_log_scale[p * C + cl]; BCme[p] = a.ch.batch_count[p * C + cl]; }
  }
  const StateView S{Sme};
  uint64_t perm = wide_perm ? 0ull : a.ch.perm[cl];
  if (wide_perm)
    for (int k = 0; k < n_named; ++k) pcol.set(k, (int)a.ch.perm16[(int64_t)k * C + cl]);
  typename RngOf<Model, G, GL, SW>::type rng;
  if constexpr (GL) rng.init(a.seed, a.chain_offset + (uint64_t)cl, a.ch.rng_n[cl], tid, reinterpret_cast<double *>(smem + L.data + Model::gl_lds_bytes(a.d.pad, 0)) + (size_t)(tid >> 6) * 256);
  else if constexpr (SW) rng.init(a.seed, a.chain_offset + (uint64_t)cl, a.ch.rng_n[cl], tid, reinterpret_cast<double *>(smem + L.data + Model::window_offset(a.d, nt / 64)) + (size_t)(tid >> 6) * 256);
  else rng.init(a.seed, a.chain_offset + (uint64_t)cl, a.ch.rng_n[cl], tid);
  double lp_curr = a.ch.lp_curr[cl];
  __syncthreads();
  // (CT: certified decisions are a kernel of their own -- amwg_step_kernel_cert / amwg_sweep_kernel_cert below --, not a run-time switch: the kernel that
  // evaluates the expression in every update carries none of the certified paths, and the certified sweep kernel none of the lane-order sums)
  constexpr bool kCert = CT && CertifiedAt<Model, G>::value && !GL && (SW == CertNeedsRows<Model>::value);      // (certified decisions, below.  One lane per chain: the Normal family; 16 lanes: the Poisson family; the sweep kernel: the hierarchical family)
  static_assert(kCert == CT, "a certified kernel is instantiated for a model that has no certified value at this lane count");
  // THE EXPRESSION: log_post of the state as it stands, term by term.  In the lane order of this geometry (log_post above: G per-lane sums and a butterfly) -- or, in
  // the certified kernels of models that can (Model::reference_order), as ONE running sum in the reference's own order: slow (every term passes through a
  // cross-lane broadcast), but it runs for ~1e-7 of the updates, and it makes what these kernels decide and store the REFERENCE's at every lane count.
  constexpr bool kRefOrder = kCert && G > 1 && RefOrderOf<Model>::value;
  auto expression = [&]() -> double {
    if constexpr (kRefOrder) return Model::template reference_order<G>(cache, S, a.mc, a.d, data_lds, sub);
    else if constexpr (!GL) return log_post<Model, G, kPassU>(S, a, data_lds, sub, xw, cache);
    else return 0.0;      // (the group-local kernel forms log_post from its pieces)
  };
  if constexpr (!GL) { if (a.init_lp) lp_curr = expression(); }  // ctor warm-up call, mcmc.js:961-963 (the group-local kernel forms it from its pieces below, in every launch)
  // CERTIFIED DECISIONS (models with Model::log_post_approx; one lane per chain).  The accept test Math.exp(prop_lp - lp_curr) > u (mcmc.js:527-528) needs the
  // two values only as far as they decide the comparison.  The model hands back a cheaper value of log_post together with a bound on its distance from what
  // the reference's expression gives (the Normal family: prior + n c - sum (x - mu)^2 / den, two operations per observation instead of eight); the
  // stepper keeps such a pair (lpA, epsA) for the current state as well, and
  //     exp(dA) (1 - eta) > u  proves the expression's own test true,   exp(dA) (1 + eta) < u  proves it false      (dA = difference of the cheap values, eta
  //     = 1.0625 (their bounds + the subtraction's rounding) + 2^-49: V8's exp is within an ulp of exp) --
  // the same argument as the 1 + d <= exp(d) bounds below.  A uniform inside the sliver (~1e-7 of the updates at cfg2) gets the expression itself, for
  // both states if need be; and whatever is stored, returned or compared -- lp_curr at the end of every launch, the ctor's value -- is the expression's.
  // Every decision, hence every draw, is the one the term-by-term evaluation makes (tests: the reference goldens, bit for bit, and the same sampler with
  // options.full_evaluation = 1, which evaluates the expression in every update).
  // (between launches the pair travels in ChainArrays::lp_curr / lp_eps: a launch does not close with an evaluation of the expression unless the host
  // asks for its value -- StepArgs::finalize_lp, amwg_chain_diag)
  double lpA = lp_curr, epsA = 0.0;      // the cheap value of log_post(current state) and its bound (0: lp_curr itself)
  bool lp_exact = true;                  // lp_curr is the expression's value of the current state
  if constexpr (kCert) { if (!a.init_lp) { epsA = a.ch.lp_eps[cl]; lp_exact = epsA == 0.0; } }
  (void)lpA; (void)epsA; (void)lp_exact;
  // BOUND AUDIT (-DAMWG_AUDIT: libamwg_audit.so, tools/bound_audit.py -- never the product).  The certified kernels decide from a cheap value A of log_post and a
  // hand-derived bound eps on |A - E|, E the expression's value; the campaigns of round 5 could not falsify a bound that is merely too small by a factor (the
  // actual |A - E| is orders of magnitude below eps, so a wrong verdict needs a uniform in a ~1e-12-wide window).  The audit build therefore evaluates E next to
  // A in EVERY audited update -- for the proposal, and carried along for the current state -- and keeps, per chain, max |A - E| / eps and max |dA - dE| / eta
  // (dE = RN(E_prop - E_cur): the difference the reference's own test takes the exponential of), the number of audited decisions, and the number of certified
  // verdicts that contradict exp_v8(dE) > u; plus histograms of the two ratios by binary exponent.  Ratios are of the bounds BEFORE test_bound_shift scales them.
#if defined(AMWG_AUDIT)
  constexpr bool kAudit = kCert;
#else
  constexpr bool kAudit = false;
#endif
  [[maybe_unused]] double audE = 0.0, aud_max_v = 0.0, aud_max_d = 0.0;      // E of the current state; the two maxima
  [[maybe_unused]] uint32_t aud_n = 0u, aud_bad = 0u;
  [[maybe_unused]] auto audit_bin = [&](int which, double ratio) {
#if defined(AMWG_AUDIT)
    if (writer) {
      int b = 0;
      if (ratio > 0.0) { b = (int)((f64_bits(ratio) >> 52) & 0x7ffu) - 1023 + 40; b = b < 1 ? 1 : (b > 63 ? 63 : b); }      // (bin 0: the two values agree exactly)
      if (!(ratio == ratio)) b = 63;
      (void)atomicAdd(cold_args()->ch.audit_hist + which * 64 + b, 1ull);
    }
#endif
  };
  // (--shrink's adversarial uniforms: u just outside the sliver of the certified test, where a bound that is too small by more than the placement's 1.5 gives a wrong verdict)
  [[maybe_unused]] auto audit_adversarial_u = [&](double ex, double eta, double u, int parity) -> double {
    if (!(eta < 0x1p-8) || !(ex > 0x1p-900) || !(ex < kInf)) return u;
    const double below = ex * (1.0 - 1.5 * eta), above = ex * (1.0 + 1.5 * eta);
    const double v = ((parity & 1) && above < 1.0) ? above : below;
    return (v > 0.0 && v < 1.0) ? v : u;
  };
  [[maybe_unused]] auto audit_value = [&](double A, double eps, double E) {      // a cheap value against the expression's
    if (eps == eps && eps < kInf && eps > 0.0 && E == E && __builtin_fabs(E) < kInf) {
      const double r = __builtin_fabs(A - E) / eps;
      aud_max_v = (r > aud_max_v || !(r == r)) ? r : aud_max_v;
      audit_bin(0, r);
    }
  };
  [[maybe_unused]] auto audit_difference = [&](double dA, double eta0, double dE, int verdict, double u) {      // ... a difference, and the verdict drawn from it
    if (eta0 == eta0 && eta0 < kInf && eta0 > 0.0 && dE == dE && __builtin_fabs(dE) < kInf) {
      const double r = __builtin_fabs(dA - dE) / eta0;
      aud_max_d = (r > aud_max_d || !(r == r)) ? r : aud_max_d;
      audit_bin(1, r);
    }
    const bool exact_acc = exp_v8(dE) > u;
    ++aud_n;
    if ((verdict > 0 && !exact_acc) || (verdict < 0 && exact_acc)) ++aud_bad;
  };
  // PHASE CLOCK (-DAMWG_X_PHASES, a development build: tools/phase_clock.py): shader-clock cycles per phase of the step loop, summed over the launch's wavefronts into
  // ChainArrays::audit_hist[64 ..] -- where a latency-bound stepper spends its time, which counters of issued instructions cannot say
#if defined(AMWG_X_PHASES)
  uint64_t ph_acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) ph_acc[q] = 0;
  uint64_t ph_last = __builtin_readcyclecounter();
#define AMWG_PHASE(i) do { const uint64_t ph_now = __builtin_readcyclecounter(); ph_acc[i] += ph_now - ph_last; ph_last = ph_now; } while (0)
#else
#define AMWG_PHASE(i) do { } while (0)
#endif
  // (set below, once `expression` and the state are usable: audE = E of the state the launch starts from)
  // every store to the state goes through here: the LDS copy (what translated closures, gathers and the final write-back read) and,
  // for models that mirror the state in registers, the mirror
  auto set_state = [&](int comp, double v) {
    Sme[comp] = v;
    if constexpr (TracksState<Model>::value) Model::on_set(cache, comp, v, sub, a.d);
  };

  constexpr int D = Model::kDerived;
  const int PR = P + D;   // recorded values per draw: the parameters, then the closure's derived quantities
  int64_t row = a.row0;
  int32_t next_rec = (int32_t)a.step0;  // host passes steps-until-first-recorded-step here
  const int P_stepped = a.pl.P_stepped;  // < P when the state has entries this sampler only reads (AMWG_FIXED)
  const bool recording = a.draws != nullptr;
  const int n_steps = a.n_steps;

  // order of the top-level entries of the multidimensional parameter being walked (mcmc.js:244-263): a column of the LDS index table, or
  // -- a chain on a whole wave and no leading dimension beyond 64 -- one entry per lane of the wave (`ord`), shuffled and read with
  // v_readlane / selects: no LDS round trip in the shuffle's 31 dependent swaps nor in the per-slot lookup
  const int lane64 = tid & 63;
  int ord = lane64;
  (void)ord;      // (the group-local kernel keeps its own order)
  // (the sweep kernel is only launched for a parameter vector of at most 64 entries -- hier_rows_wanted, amwg_core.hip --: the order is always in
  // registers there, and the LDS-table walk is not compiled in; a launch that breaks this is refused like the other device-side invariants)
  if (SW && a.pl.max_top > 64) { device_error(a, kErrSweepNeedsOrderInRegisters); return; }
  const bool ord_in_regs = SW ? true : (G >= 64 && a.pl.max_top <= 64);
  // sweep prefetch (below): a model with the lane-local re-evaluation in its row layout, a chain on one whole wavefront, the order in registers
  // (SW: its own kernel, amwg_sweep_kernel below -- the host launches it when the row layout is in use; the ordinary kernel, which is also the
  // full-evaluation one, does not carry the extra registers: with the sweep compiled into it, it ran a third slower)
  constexpr bool kSweep = SW && !GL && LaneReuseOf<Model>::value && G == 64 && !BinaryOf<Model>::value;
  const bool sweep_rt = kSweep && ord_in_regs && a.d.pad > 0;
  if constexpr (kAudit) { if (lp_exact) audE = lp_curr; else audE = expression(); }      // (BOUND AUDIT: the expression's value of the state the launch starts from)
  wave_priority(kStepperPriority);
  // ---- Sampler.sample: record the state BEFORE the step (mcmc.js:1020-1027)
  auto record_draws = [&](int step) {
    if (recording && step == next_rec) {
      double *const draws = cold_args()->draws;
      // the row is written by ALL lanes of the chain that sit in its first wavefront -- lane j components j, j + L, ... (L = min(G, 64)): one store instruction per L
      // components.  (Round 5 had the chain's first lane walk the P components: 34 dependent single-lane stores per step of cfg4, a quarter of a recording launch.)
      {
        constexpr int LW = G < 64 ? G : 64;
        if (live && (!kMulti || sub < 64))
          for (int p = sub & (LW - 1); p < P; p += LW) {
#if !defined(AMWG_X_NOREC)      // (development experiment: a recording launch without its stores, tools/build_variant.sh)
            draws[(row * PR + p) * C + cl] = S(p);
#endif
          }
      }
      if constexpr (D > 0) {
        // derived quantities (`state.var = ...` inside log_post, mcmc.js:961-963, 990-995): the
        // reference re-evaluates log_post at the end of every step, so what sample() records is the
        // closure's assignment at the recorded state.
        double dv[D];
        (void)Model::template eval<G, true>(S, a.d, data_lds, sub, dv);
        if (writer)
          for (int q = 0; q < D; ++q) draws[(row * PR + P + q) * C + cl] = dv[q];
      }
      ++row;
      next_rec += cold_args()->thin;
    }
  };
  // ---- AmwgStepper.step: in-place Durstenfeld shuffle of the named sub-steppers (mcmc.js:887, 228-236)
  auto shuffle_named = [&]() {
    for (int i = n_named - 1; i > 0; --i) {
      const int j = (int)__builtin_floor(rng.next() * (double)(i + 1));
      if (wide_perm) { const int ti = pcol.get(i); pcol.set(i, pcol.get(j)); pcol.set(j, ti); }
      else perm = perm_swap(perm, i, j);
    }
  };
  // ---- Roberts-Rosenthal batch adaptation of one component (mcmc.js:536-550); `store`: this lane writes the component's HBM words
  auto adapt_component = [&](int comp, bool accepted, int2 cnt, double batch_size, bool store, bool per_lane = false) {
    cnt.x += accepted ? 1 : 0;      // acceptance_count (mcmc.js:530)
    cnt.y += 1;                     // iterations_since_adaption (mcmc.js:537)
    const bool boundary = (double)cnt.y >= batch_size;
    if (per_lane ? boundary : chain_true<GL ? 1 : G>(boundary)) {    // batch boundary: the only time batch_count is touched (it stays in HBM); (the group-local sweep calls this per lane)
      const CompConst k = cc[comp];
      // single-wave chains: batch_count and the log scale live in HBM (all lanes of the chain are in lockstep, so they
      // read the old value together before the writer lane stores the new one); multi-wave chains keep per-wave replicas
      const int64_t gi = (int64_t)comp * C + cl;
      int32_t *const g_bc = kMulti ? nullptr : cold_args()->ch.batch_count;
      double *const g_pls = kMulti ? nullptr : cold_args()->ch.prop_log_scale;
      const int32_t bc = (kMulti ? BCme[comp] : g_bc[gi]) + 1;
      const double adj = __builtin_fmin(k.max_adaptation, k.initial_adaptation / __builtin_sqrt((double)bc));
      double pls = kMulti ? LOGPLSme[comp] : g_pls[gi];
      if ((double)cnt.x / k.batch_size > k.target_accept_rate) pls += adj; else pls -= adj;
      cnt = make_int2(0, 0);
      SDme[comp] = exp_v8_cold(pls);
      if constexpr (kMulti) { BCme[comp] = bc; LOGPLSme[comp] = pls; }
      else if (store) { g_bc[gi] = bc; g_pls[gi] = pls; }
    }
    CNTme[comp] = cnt;
  };

  // Math.exp(prop - curr) > Math.random() (mcmc.js:527-528), for a chain whose lanes all hold the same difference and uniform.  For a difference
  // >= 0 (incl. +inf) the exponential is >= 1 > u, below -746 it is exactly 0 (never > u): the decision is the reference's without evaluating it; NaN
  // takes the general path (false).  For d < 0:  1 + d <= exp(d) <= 1 + d + d*d/2, and V8's exp is within one ulp (< 2^-53 here) of exp: a uniform below
  // the lower bound or above the upper one (each taken with a margin of 2^-50, an order of magnitude more than the roundings of the bounds themselves
  // plus that ulp) decides the comparison exactly as the exponential would -- which is then evaluated for the band in between only (about a quarter of
  // the proposals at a 44 % acceptance rate).
  [[maybe_unused]] auto accept_sweep = [&](double diff, double u_accept) -> bool {
    bool accepted = false;
    if (chain_true<G>(diff >= 0.0)) accepted = true;
    else if (chain_true<G>(diff < -746.0)) accepted = false;
    else {
      const double lower = 1.0 + diff;
      if (chain_true<G>(u_accept < lower - 0x1p-50)) accepted = true;
      else if (chain_true<G>(diff > -1.0 && u_accept > (lower + 0.5 * diff * diff) + 0x1p-50)) accepted = false;
      else accepted = chain_true<G>(exp_v8(diff) > u_accept);
    }
    return chain_true<G>(accepted);
  };

  // ================================================================================================================================
  // GROUP-LOCAL kernel (GL; amwg_gl.h has the method and its order of additions): the named parameters are walked in the shuffled order
  // as always, mu and sigma by the ordinary stepper with the group-local evaluation, and the Gn components of theta in ONE lane-parallel
  // sweep.  Same uniforms for the same purposes in the same order as the sequential stepper (oracle: gl_evaluate).
#ifndef AMWG_X_GLCUT
#define AMWG_X_GLCUT 0      // development experiments (wrong results): leave one piece of the group-local step out, to price it (a build with -DAMWG_X_GLCUT=n: tools/build_variant.sh)
#endif
  if constexpr (GL) {
    static_assert(G == 64, "the group-local kernel runs a chain on one whole wavefront");
    typename Model::Lane gl;
    const int Gn = a.d.G;
    lp_curr = Model::template refresh<kPassU>(gl, S, a.mc, a.d, data_lds, lane64);
    const int comp_l = gl.grp >= 0 ? gl.grp : 0;            // this lane's component of theta (theta is the first parameter); idle lanes: masked
    const int first_of = (int)reinterpret_cast<const int8_t *>(data_lds + (size_t)a.d.pad * 64 * 8)[lane64 * 8 + 7];      // GlLane::first_of of lane c: the first lane of component c's block
    uint64_t first_mask = __ballot(gl.first);
    for (int step = 0; step < n_steps; ++step) {
      record_draws(step);
      shuffle_named();
      for (int np = 0; np < n_named; ++np) {
        const int p = __builtin_amdgcn_readfirstlane((int)perm_get(perm, np));
        const int base = __builtin_amdgcn_readfirstlane(pl_base[p]);
        if (__builtin_amdgcn_readfirstlane(pl_multidim[p]) == 0) {
          // ---- mu or sigma: OnedimMetropolisStepper.step (mcmc.js:517-553) with the group-local evaluation
          if (AMWG_X_GLCUT == 6) continue;
          const int comp = base;
          const bool is_mu = comp == Gn;
          const double cur = is_mu ? gl.mu : gl.sigma;
          const double k_lower = cc[comp].lower, k_upper = cc[comp].upper;
          const int k_type = cc[comp].type;
          const int2 cnt = CNTme[comp];
          const double batch_size = cc[comp].batch_size;
          const bool adapting = adapt[comp] != 0;
          double prop = rnorm_js(rng, cur, SDme[comp]);
          if (k_type == kTypeInt) prop = js_round(prop);
          const bool inb = !(prop < k_lower || prop > k_upper);
          bool accepted = false;
          if (inb) {
            const double u_accept = rng.next();
            const double prop_lp = Model::template eval_scalar<kPassU>(gl, is_mu, prop, a.mc, a.d, data_lds, lane64);
            // Math.exp(diff) > u (mcmc.js:527-528), decided without the exponential where 1 + d <= exp(d) <= 1 + d + d*d/2 (d < 0) already tells (see the
            // ordinary stepper below): every lane of the wavefront holds the same chain, so skipping it here really skips it
            const double diff = prop_lp - lp_curr;
            if (diff >= 0.0) accepted = true;
            else if (diff < -746.0) accepted = false;
            else {
              const double lower = 1.0 + diff;
              if (u_accept < lower - 0x1p-50) accepted = true;
              else if (diff > -1.0 && u_accept > (lower + 0.5 * diff * diff) + 0x1p-50) accepted = false;
              else accepted = exp_v8(diff) > u_accept;
            }
            if (accepted) { lp_curr = prop_lp; Model::commit_scalar(gl, is_mu, prop); Sme[comp] = prop; }
            if (counter) (void)__hip_atomic_fetch_add(&TOTme[comp], 1u + (accepted ? 0x10000u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          if (adapting) adapt_component(comp, accepted, cnt, batch_size, writer);
          continue;
        }
        // ---- theta.  A fresh shuffle of the order (mcmc.js:248-252): the Fisher-Yates loop i = top-1 .. 1 draws one uniform per round, i.e. the
        // next top - 1 uniforms of the stream -- lane i fetches and scales ITS one, then the transpositions (i, j_i) are applied in sequence to
        // every lane's POSITION in the order (lane c < top tracks where component c sits).
        if (AMWG_X_GLCUT == 9) continue;
        const int top = Gn;
        uint32_t p_s = rng.position();                          // < 128: once its second half is there the window reaches at least 128 uniforms ahead
        rng.ensure_b();
        int jv = 0;
        {
          const uint32_t q = p_s + (uint32_t)(top - 1 - lane64);      // round i = lane consumes uniform number (top - 1 - i) of the shuffle
          const double u = rng.at(lane64 >= 1 && lane64 < top ? q : 0u);
          jv = (int)__builtin_floor(u * (double)(lane64 + 1));
        }
        int posc = lane64;
        if (AMWG_X_GLCUT != 1) {
          // (four transpositions per trip, their j's read ahead of the dependent selects: the chain posc -> compare -> select is all that is serial)
          int i = top - 1;
          for (; i >= 4; i -= 4) {
            const int j0 = __builtin_amdgcn_readlane(jv, i), j1 = __builtin_amdgcn_readlane(jv, i - 1), j2 = __builtin_amdgcn_readlane(jv, i - 2), j3 = __builtin_amdgcn_readlane(jv, i - 3);
            posc = posc == i ? j0 : (posc == j0 ? i : posc);
            posc = posc == i - 1 ? j1 : (posc == j1 ? i - 1 : posc);
            posc = posc == i - 2 ? j2 : (posc == j2 ? i - 2 : posc);
            posc = posc == i - 3 ? j3 : (posc == j3 ? i - 3 : posc);
          }
          for (; i > 0; --i) {
            const int j = __builtin_amdgcn_readlane(jv, i);
            posc = posc == i ? j : (posc == j ? i : posc);
          }
        }
        p_s += (uint32_t)(top - 1);
        // my component's place in the order, and -- lane t -- the component at place t
        const int pos_l = __shfl(posc, comp_l, 64);
        const int ordv = __builtin_amdgcn_ds_permute(posc << 2, lane64);      // lane posc[c] receives c (lanes >= top map onto themselves)
        const double sd_l = SDme[comp_l], lower_l = cc[comp_l].lower, upper_l = cc[comp_l].upper, bs_l = cc[comp_l].batch_size;
        const int type_l = cc[comp_l].type;
        const bool adapting_l = adapt[comp_l] != 0;
        const bool bounded_any = __ballot(gl.grp >= 0 && (lower_l > -kInf || upper_l < kInf || type_l == kTypeInt)) != 0ull;
        uint64_t inb_assume = ~0ull;       // (bit = first lane of a component's block) which proposals are assumed to fall inside their bounds: they draw the accept uniform
        int ppv = 0;                       // lane t: the stream position of the accepted (u, v) pair of update t of the order
        int t_begin = 0;
        while (t_begin < top) {
          rng.ensure_b();
          // -- scalar resolution: update t of the order takes the first accepted pair at or after the stream position, then (if its proposal is
          // inside the bounds) one more uniform for the accept test.  A pair must start at or before 253 (its accept uniform is inside the window).
          // (every input of the loop is handed over as a scalar: the compiler then keeps the loop itself on the scalar unit)
          const uint64_t EA = uniform_u64(rng.EA), OA = uniform_u64(rng.OA), EB = uniform_u64(rng.EB) & ~(1ull << 63), OB = uniform_u64(rng.OB);
          const uint64_t assume = uniform_u64(inb_assume);
          const bool bounded = __builtin_amdgcn_readfirstlane((int)bounded_any) != 0;
          const int top_s = __builtin_amdgcn_readfirstlane(top);
          uint32_t p = (uint32_t)__builtin_amdgcn_readfirstlane((int)p_s);
          int t = __builtin_amdgcn_readfirstlane(t_begin);
          if (!bounded && AMWG_X_GLCUT != 2) {
            // every proposal draws its accept uniform (no bounds): the loop as ~17 scalar instructions + one v_writelane per update.  (The
            // compiler's version of the general loop below kept its flags in vector registers and branched a dozen times per update: 16 vector +
            // 30 scalar instructions per update, a sixth of the whole step.)
            gl_resolve_unbounded(EA, OA, EB, OB, p, t, top_s, ppv);
          } else
          for (; t < top_s; ++t) {
            bool found = AMWG_X_GLCUT == 2 && p < 250u;
            while (AMWG_X_GLCUT != 2 && p < 256u) {
              const uint32_t par = p & 1u, idx = (p & 127u) >> 1;
              const uint64_t m = (p & 128u) ? (par ? OB : EB) : (par ? OA : EA);
              const uint64_t rest = m >> idx;
              if (rest != 0ull) { p += 2u * (uint32_t)__builtin_ctzll(rest); found = true; break; }
              if (p & 128u) { p = 254u | par; break; }      // every pair up to the window's last one is rejected (consumed): the search resumes there in the next window
              p = 128u | par;                               // nothing left in the first half: on to the second
            }
            if (!found) break;
            ppv = write_lane(ppv, (int)p, t);
            const int ck = __builtin_amdgcn_readlane(ordv, t);
            const int fl = __builtin_amdgcn_readlane(first_of, ck);
            p += ((assume >> fl) & 1ull) ? 3u : 2u;
          }
          const int t_end = t;
          // -- every lane: the proposal of its own group
          const bool in_round = gl.grp >= 0 && pos_l >= t_begin && pos_l < t_end;
          // (the shuffle is done by ALL lanes, before the select: inside a conditional expression it runs with the other lanes masked off, and a lane
          // that reads from a masked-off lane gets 0 -- which is what a sweep that needs a second window then read)
          int q_raw = __shfl(ppv, pos_l, 64);
          asm volatile("" : "+v"(q_raw));
          const uint32_t q_l = in_round ? (uint32_t)q_raw : 0u;
          const double u = rng.at(q_l), v_raw = rng.at(q_l + 1u), u_accept = rng.at(q_l + 2u);
          const double cur = gl.th;
          double prop = ((1.7156 * (v_raw - 0.5)) / u) * sd_l + cur;       // rnorm_js: (v / u) * sd + mean
          if (type_l == kTypeInt) prop = js_round(prop);
          const bool inb = !(prop < lower_l || prop > upper_l);
          if (bounded_any) {
            // the resolution assumed which proposals draw an accept uniform: check, and resolve again where it was wrong (each pass fixes at
            // least the earliest wrong one; with unbounded components there is nothing to fix)
            const uint64_t round_mask = __ballot(in_round) & first_mask, actual = __ballot(inb) & first_mask;
            if (((actual ^ inb_assume) & round_mask) != 0ull) { inb_assume = (inb_assume & ~round_mask) | (actual & round_mask); continue; }
          }
          const bool eval = in_round && inb;
          double Ls_t;
          const double delta = Model::template sweep_eval<kPassU>(gl, eval, prop, Ls_t, a.mc, a.d, data_lds, lane64);
          // Math.exp(delta) > u (mcmc.js:527-528); >= 0 and < -746 decided without the exponential, NaN takes it and fails
          bool accepted = false;
          if (eval) {
            if (delta >= 0.0) accepted = true;
            else if (delta < -746.0) accepted = false;
            else accepted = AMWG_X_GLCUT == 4 ? false : exp_v8(delta) > u_accept;
          }
          Model::sweep_commit(gl, accepted, prop, Ls_t);
          if (AMWG_X_GLCUT != 8 && gl.first && in_round) {            // one lane per component: state, run totals, adaptation
            if (accepted) Sme[comp_l] = prop;
            if (inb) TOTme[comp_l] += 1u + (accepted ? 0x10000u : 0u);
            if (adapting_l) adapt_component(comp_l, accepted, CNTme[comp_l], bs_l, live);
          }
          t_begin = t_end;
          // the stream position after this round (>= 128: the window moves on when it is next looked at)
          p_s = p;
          rng.pos = p;
          if (t_begin < top) p_s = rng.position();
        }
        lp_curr = Model::total(gl, lane64, gl.pm, gl.pt, gl.T);
      }
    }
  } else
  for (int step = 0; step < n_steps; ++step) {
    AMWG_PHASE(15);
    record_draws(step);
    shuffle_named();
    AMWG_PHASE(0);
    // ---- every scalar component exactly once; `slot` is uniform across the block
    int np = 0, e = 0, e_top = 0, e_in = 0;   // position inside the current parameter: e = e_top * inner + e_in (no division per slot)
    // Sweep prefetch (models with the lane-local re-evaluation, Model::prefetch_rows): when the walk reaches the vector parameter whose entries are
    // the lanes' group means, the proposals and accept uniforms of ALL its updates are drawn at once -- the same uniforms for the same purposes in
    // the same order: nothing an update draws depends on an earlier decision -- and kept one per lane (lane c: component c); the model forms
    // every lane's proposed sum in one pass, and the updates then run as always, taking their proposal and uniform from the lanes.
    double sw_prop = 0.0, sw_u = 0.0;
    int ord_pos = lane64;     // lane c: the place of entry c in the shuffled order (the inverse of `ord`; sweep kernel only)
    (void)ord_pos;
    int sw_left = 0;          // updates of the sweep still to come (0: the stepper draws as it goes)
    int sb = 0;               // the swept vector's place in the state: its entry c is component sb + c (the built-in family: 0)
    if constexpr (kSweep) sb = Model::sweep_base(a.d);
    bool sw_pending = false;  // the parameter whose first slot comes next is such a sweep: drawn at the top of that slot
    // descriptor of the parameter being walked, read from the LDS tables once, when the parameter begins (round 2 re-read it in every
    // slot: a dependent LDS round trip per update in front of the component lookup)
    int d_len = 1, d_base = 0, d_multi = 0, d_inner = 1;
    // the component the next slot updates.  Performs the fresh shuffle when a multidimensional parameter begins (mcmc.js:248-252), which
    // consumes uniforms: it must be called in stream order, i.e. after everything the previous slot draws.
    auto next_comp = [&]() -> int {
      if (e == 0) {
        const int p = chain_uniform<G>(wide_perm ? pcol.get(np) : (int)perm_get(perm, np));
        d_len = chain_uniform<G>(pl_len[p]);
        d_base = chain_uniform<G>(pl_base[p]);
        d_multi = chain_uniform<G>(pl_multidim[p]);
        if (d_multi) {
          const int top = chain_uniform<G>(pl_top[p]);
          d_inner = chain_uniform<G>(pl_inner[p]);
          if (ord_in_regs) {
#if defined(__HIP_DEVICE_COMPILE__)
            if constexpr (kSweep) {
              // the window stream: round i = top-1 .. 1 of the Fisher-Yates loop draws uniform number top-1-i of the next top-1 -- lane i fetches
              // and scales ITS one, then the transpositions (i, j_i) are applied in sequence to every lane's POSITION (lane c tracks where entry c
              // sits); four per trip, their j's read ahead of the dependent selects (as in the group-local kernel)
              const uint32_t p0 = rng.position();
              rng.ensure_b();
              const double uf = rng.at(lane64 >= 1 && lane64 < top ? p0 + (uint32_t)(top - 1 - lane64) : 0u);
              const int jv = (int)__builtin_floor(uf * (double)(lane64 + 1));
              int posc = lane64, i = top - 1;
              for (; i >= 4; i -= 4) {
                const int j0 = __builtin_amdgcn_readlane(jv, i), j1 = __builtin_amdgcn_readlane(jv, i - 1), j2 = __builtin_amdgcn_readlane(jv, i - 2), j3 = __builtin_amdgcn_readlane(jv, i - 3);
                posc = posc == i ? j0 : (posc == j0 ? i : posc);
                posc = posc == i - 1 ? j1 : (posc == j1 ? i - 1 : posc);
                posc = posc == i - 2 ? j2 : (posc == j2 ? i - 2 : posc);
                posc = posc == i - 3 ? j3 : (posc == j3 ? i - 3 : posc);
              }
              for (; i > 0; --i) {
                const int j = __builtin_amdgcn_readlane(jv, i);
                posc = posc == i ? j : (posc == j ? i : posc);
              }
              rng.pos = p0 + (uint32_t)(top - 1);
              ord_pos = posc;
              ord = __builtin_amdgcn_ds_permute(posc << 2, lane64);      // lane posc[c] receives c (lanes >= top map onto themselves)
            } else if constexpr (G >= 64) {
              ord = lane64;
              for (int i = top - 1; i > 0; --i) {
                const int j = __builtin_amdgcn_readfirstlane((int)__builtin_floor(rng.next() * (double)(i + 1)));
                const int ti = __builtin_amdgcn_readlane(ord, i), tj = __builtin_amdgcn_readlane(ord, j);
                ord = lane64 == i ? tj : (lane64 == j ? ti : ord);
              }
            }
#endif
          } else {
            for (int t = 0; t < top; ++t) idx.set(t, t);
            for (int i = top - 1; i > 0; --i) {
              const int j = (int)__builtin_floor(rng.next() * (double)(i + 1));
              const int ti = idx.get(i);
              idx.set(i, idx.get(j));
              idx.set(j, ti);
            }
          }
          if constexpr (kSweep) sw_pending = sweep_rt && d_base == Model::sweep_base(a.d) && d_inner == 1 && d_len == chain_uniform<G>(Model::sweep_len(a.d)) && d_len > 1;
        }
      }
      int comp = d_base;
      if (d_multi) {
        int t = 0;
        if (ord_in_regs) {
#if defined(__HIP_DEVICE_COMPILE__)
          if constexpr (G >= 64) t = __builtin_amdgcn_readlane(ord, __builtin_amdgcn_readfirstlane(e_top));
#endif
        } else t = idx.get(e_top);
        comp += t * d_inner + e_in;
        if (++e_in == d_inner) { e_in = 0; ++e_top; }
      }
      if (++e == d_len) { e = 0; e_top = 0; e_in = 0; ++np; }
      return comp;
    };
    auto prefetch = [&](int comp) -> SlotPre {
      SlotPre q;
      q.comp = comp;
      q.cur = Sme[comp];
      q.sd = SDme[comp];
      q.cnt = CNTme[comp];
      q.batch_size = cc[comp].batch_size;
      q.adapting = adapt[comp] != 0;
      return q;
    };
    SlotPre nx{};
    if (P_stepped > 0) nx = prefetch(next_comp());
    AMWG_PHASE(1);
    for (int slot = 0; slot < P_stepped; ++slot) {
      if constexpr (kSweep) {
        if (sw_pending) {      // (wave-uniform) the first slot of the sweep: draw everything its updates draw, in their order
          sw_pending = false;
          AMWG_PHASE(14);
#if defined(__HIP_DEVICE_COMPILE__)
          // Every lane draws the proposal of ITS component (sweep_comp: the one its sum depends on).  Which (u, v) pair of the stream survives rnorm's
          // rejection test (mcmc.js:44-53) is a property of the stream alone: the window stream holds those flags for 256 uniforms, a scalar loop walks
          // the updates in their shuffled order -- first accepted pair at or after the position, then the accept uniform (as in the group-local kernel,
          // gl_resolve_unbounded) -- and a lane reads the three uniforms of its component's update from the window.  An integer component rounds its
          // proposal (mcmc.js:597), which draws nothing.  A component with BOUNDS draws no accept uniform when its proposal falls outside (mcmc.js:520-522):
          // the walk assumes every proposal inside, the lanes check their own, and where the assumption was wrong the round is walked again with it
          // corrected (the earliest wrong one is final after each pass, as in the group-local kernel) -- round 4 walked such a parameter update by update.
          const int comp_l = Model::sweep_comp(data_lds, a.d, sub), cl = comp_l >= 0 ? comp_l : 0, cidx = sb + cl;
          const double sd_l = SDme[cidx], cur_l = Sme[cidx];
          const double lower_l = cc[cidx].lower, upper_l = cc[cidx].upper;
          const int type_l = cc[cidx].type;
          const bool bounded_any = __ballot(comp_l >= 0 && (lower_l > -kInf || upper_l < kInf)) != 0ull;
          uint64_t inb_assume = ~0ull;      // bit c: the proposal of entry c is taken to fall inside its bounds (it draws the accept uniform)
          {
          {
            const int top = d_len;
            const int pos_l = __shfl(ord_pos, cl, 64);
            uint32_t p_s = rng.position();
            int ppv = 0, t_begin = 0;
            while (t_begin < top) {
              rng.ensure_b();
              AMWG_PHASE(13);
              const uint64_t EA = uniform_u64(rng.EA), OA = uniform_u64(rng.OA), EB = uniform_u64(rng.EB) & ~(1ull << 63), OB = uniform_u64(rng.OB);
              uint32_t pw = (uint32_t)__builtin_amdgcn_readfirstlane((int)p_s);
              int t = __builtin_amdgcn_readfirstlane(t_begin);
              const int top_s = __builtin_amdgcn_readfirstlane(top);
              if (!bounded_any) gl_resolve_unbounded(EA, OA, EB, OB, pw, t, top_s, ppv);
              else {
                const uint64_t assume = uniform_u64(inb_assume);
                for (; t < top_s; ++t) {
                  bool found = false;
                  while (pw < 256u) {
                    const uint32_t par = pw & 1u, ix = (pw & 127u) >> 1;
                    const uint64_t m = (pw & 128u) ? (par ? OB : EB) : (par ? OA : EA);
                    const uint64_t rest = m >> ix;
                    if (rest != 0ull) { pw += 2u * (uint32_t)__builtin_ctzll(rest); found = true; break; }
                    if (pw & 128u) { pw = 254u | par; break; }      // every pair up to the window's last one is rejected (consumed): the search resumes there in the next window
                    pw = 128u | par;                               // nothing left in the first half: on to the second
                  }
                  if (!found) break;
                  ppv = write_lane(ppv, (int)pw, t);
                  const int ck = __builtin_amdgcn_readlane(ord, t);
                  pw += ((assume >> ck) & 1ull) ? 3u : 2u;
                }
              }
              const int t_end = t;
              const bool in_round = comp_l >= 0 && pos_l >= t_begin && pos_l < t_end;
              int q_raw = __shfl(ppv, pos_l, 64);      // (by all lanes, outside the select: a masked-off source lane reads as 0)
              asm volatile("" : "+v"(q_raw));
              const uint32_t q_l = in_round ? (uint32_t)q_raw : 0u;
              const double u = rng.at(q_l), v_raw = rng.at(q_l + 1u), ua = rng.at(q_l + 2u);
              double prop_l = ((1.7156 * (v_raw - 0.5)) / u) * sd_l + cur_l;       // rnorm_js: (v / u) * sd + mean
              if (type_l == kTypeInt) prop_l = js_round(prop_l);
              if (bounded_any) {
                const bool inb_me = !(prop_l < lower_l || prop_l > upper_l);
                const uint64_t round_mask = __ballot(in_round && lane64 < top), actual = __ballot(inb_me);      // (lane c < top: entry c)
                if (((actual ^ inb_assume) & round_mask) != 0ull) { inb_assume = (inb_assume & ~round_mask) | (actual & round_mask); continue; }
              }
              sw_prop = in_round ? prop_l : sw_prop;
              sw_u = in_round ? ua : sw_u;
              t_begin = t_end;
              p_s = pw;
              rng.pos = pw;
              if (t_begin < top) p_s = rng.position();
            }
          }
          AMWG_PHASE(2);
          const uint64_t sw_inb = (d_len >= 64 ? ~0ull : ((1ull << d_len) - 1ull)) & inb_assume;      // (a proposal outside its bounds is not evaluated)
          // (the sums are prepared for proposals INSIDE their bounds; an entry whose proposal fell outside keeps its value: its lanes' sums are not used)
          const bool inb_mine = ((inb_assume >> cl) & 1ull) != 0ull;
          const double sw_eval = (bounded_any && !inb_mine) ? cur_l : sw_prop;
          // CERTIFIED SWEEP (models with Model::sweep_approx: the hierarchical family).  Every lane's sum under its entry's proposal as  start' + n_l c - S2' / den
          // with S2' the sum of squares of its row about the proposed mean -- two operations per observation, where the prefetch below forms the
          // expression's sum in eight --, the sweep's accept tests from the entries' local differences of THOSE values with the bound that goes with them
          // (Model::difference_bound).  If every test is decided, the accepted entries' values, the state and one butterfly are all that is left to do, and the
          // stepper carries on with the cheap value of log_post and its bound; else the sweep takes the path below (the expression's sums) as if this had not run.
          bool sweep_certified = false;
          if constexpr (kCert) {
            const int top = d_len;
            if (slot + d_len <= P_stepped && (top & (top - 1)) == 0) {
              const auto sa = Model::sweep_approx(cache, S, a.mc, a.d, data_lds, sub, sw_eval);      // {ok, comp, cur, neu (values), mag, mean_new, s2_new}
              AMWG_PHASE(3);
              const bool regular = sa.ok && __ballot(sa.comp != (lane64 & (top - 1))) == 0ull;
              if (regular) {
                double dsum = sa.neu - sa.cur;
                if (top <= 32) dsum = xor_sum<32>(dsum);
                if (top <= 16) dsum = xor_sum<16>(dsum);
                if (top <= 8) dsum = xor_sum<8, true>(dsum);
                if (top <= 4) dsum = xor_sum<4, true>(dsum);
                if (top <= 2) dsum = xor_sum<2>(dsum);
                const double M = butterfly<1, 64>(sa.mag);
                const double eta = ((Model::difference_bound(M, a.d) + __builtin_fabs(dsum) * 0x1p-51) * 1.0625 + 0x1p-49) * a.bound_scale;
                AMWG_PHASE(4);
                const double ex = exp_v8(dsum);
                AMWG_PHASE(5);
                const uint64_t inb_s = uniform_u64(sw_inb);
                const bool valid = lane64 < top && ((inb_s >> lane64) & 1ull) != 0ull;
                if constexpr (kAudit) { if (a.audit_adversarial) sw_u = audit_adversarial_u(ex, eta, sw_u, step + lane64); }
                const int verdict = certified_test(dsum, eta, ex, sw_u);
                const bool sure_acc = verdict > 0;
                const bool unsure = valid && verdict == 0;
                [[maybe_unused]] double aud_E_walk = audE;
                if constexpr (kAudit) {
                  // BOUND AUDIT of the all-at-once decision: the sweep walked update by update in its order with the EXPRESSION -- entry c proposed on top of whatever the
                  // expression's own tests accepted before it, exactly the reference's schedule -- and every entry's local difference dsum_c held against RN(E_prop - E_cur).
                  // State and register mirror are put back afterwards; the walk's final E is the sweep's if the certified path then takes it.
                  const auto cache_keep = cache;
                  const double eta0 = (Model::difference_bound(M, a.d) + __builtin_fabs(dsum) * 0x1p-51) * 1.0625;
                  for (int t = 0; t < top; ++t) {
                    const int c = __builtin_amdgcn_readlane(ord, t);
                    if (!((inb_s >> c) & 1ull)) continue;
                    const double prop_c = lane_value(sw_prop, c), u_c = lane_value(sw_u, c), d_c = lane_value(dsum, c), e_c = lane_value(eta0, c);
                    const int v_c = __builtin_amdgcn_readlane(verdict, c);
                    const double old_c = Sme[sb + c];
                    set_state(sb + c, prop_c);
                    const double Ep = expression();
                    const double dE = Ep - aud_E_walk;
                    audit_difference(d_c, e_c, dE, v_c, u_c);
                    if (exp_v8(dE) > u_c) aud_E_walk = Ep; else set_state(sb + c, old_c);
                  }
                  if (lane64 < top) Sme[sb + lane64] = cur_l;
                  cache = cache_keep;
                }
                if (__ballot(unsure) == 0ull) {
                  sweep_certified = true;
                  const uint64_t acc_mask = __ballot(valid && sure_acc);
                  const bool mine = ((acc_mask >> (sa.comp & 63)) & 1ull) != 0ull;
                  if (lane64 < top && ((acc_mask >> lane64) & 1ull) != 0ull) Sme[sb + lane64] = sw_prop;
                  Model::sweep_approx_commit(cache, sa, acc_mask, mine, sw_prop, sub, a.d);
                  // the cheap value of log_post of the state the sweep leaves, and its bound: what the following updates are decided against
                  lpA = butterfly<1, 64>(mine ? sa.neu : sa.cur);
                  epsA = Model::value_bound(M, a.d);
                  lp_exact = false;
                  if constexpr (kAudit) { audE = aud_E_walk; audit_value(lpA, epsA, audE); }      // (the cheap value of the state the sweep leaves, against the walk's)
                  if (lane64 < top) {
                    const bool inb_l = ((inb_s >> lane64) & 1ull) != 0ull, acc_l = ((acc_mask >> lane64) & 1ull) != 0ull;
                    if (inb_l) TOTme[sb + lane64] += 1u + (acc_l ? 0x10000u : 0u);
                    if (adapt[sb + lane64] != 0) adapt_component(sb + lane64, acc_l, CNTme[sb + lane64], cc[sb + lane64].batch_size, live, true);
                  }
                  AMWG_PHASE(6);
                  e = 0; e_top = 0; e_in = 0; ++np;
                  slot += d_len - 1;
                  if (slot + 1 < P_stepped) nx = prefetch(next_comp());
                  AMWG_PHASE(7);
                }
              }
            }
            // (the path below compares the expression's values in this geometry's lane order: lp_curr must be one)
            if constexpr (!kRefOrder) { if (!sweep_certified && !lp_exact) { lp_curr = expression(); lp_exact = true; lpA = lp_curr; epsA = 0.0; } }
          }
          if (sweep_certified) continue;
          // (deciding against the reference's order: a sweep that is not certified as a whole is walked update by update -- each update certified on its own or
          // decided by the expression in that order; the lanes' sums in THIS geometry's order, which the path below compares, are not what is compared then)
          constexpr bool by_sums = !kRefOrder;
          using SweepRowsT = decltype(Model::template prefetch_rows<kPassU>(cache, S, a.mc, a.d, data_lds, sub, sw_eval, a.d.pad));
          SweepRowsT rows{};      // (ok = false)
          if constexpr (by_sums) rows = Model::template prefetch_rows<kPassU>(cache, S, a.mc, a.d, data_lds, sub, sw_eval, a.d.pad);
          if (by_sums && rows.ok && slot + d_len <= P_stepped) {
            // every lane holds its committed sum and its sum under its component's proposal: an update is the butterfly of the 64 sums with the
            // proposed ones in the lanes of ITS component -- the value the whole evaluation returns for that state, bit for bit --, the accept test,
            // and on acceptance the new sums, value and state; counters and adaptation afterwards, every component in its own lane
            double T_cur = rows.T_cur;
            uint64_t acc_mask = 0ull;
            const uint64_t inb_s = uniform_u64(sw_inb);
            // ALL the sweep's accept tests at once, lane c deciding entry c.  The test of an update is exp(prop_lp - lp_curr) > u with both values butterflies of
            // the 64 per-lane sums, which differ in the lanes of ITS entry only: as real numbers their difference is D_c = sum over those lanes of
            // (T_new - T_cur) whatever was accepted before (an entry is updated once per sweep, and nothing else reaches its lanes' sums).  The rounded
            // butterflies are each within 6 x 2^-53 x M of the real sums (pairwise summation over 6 levels; M = sum over the lanes of max(|T_cur|, |T_new|), a
            // bound on every vector of sums the sweep can pass through), D_c as computed here within 28 x 2^-53 x M of the real difference: the stepper's
            // difference lies within eps = 2^-46 M of D_c, its exponential within a factor 1 +- eta (eta = 1.0625 eps + 2^-49: V8's exp is within an ulp of
            // exp) of exp(D_c).  A uniform outside that sliver decides the test exactly as the update-by-update evaluation would -- the same argument as the
            // 1 + d <= exp(d) bounds of accept_sweep; if ANY entry's uniform falls inside (some 1e-8 of the sweeps at cfg4) the sweep is walked update by
            // update as before.  Afterwards the committed sums, the state and ONE butterfly give what the last accepted update would have left: the same bits.
            bool decided = false;
            {
              const int top = d_len;
              const bool regular = !a.sweep_update_by_update && (top & (top - 1)) == 0 && __ballot(rows.comp != (lane64 & (top - 1))) == 0ull;      // entry c in lanes c, c + top, ... (labels i mod top)
              if (regular) {
                double dsum = rows.T_new - rows.T_cur;
                if (top <= 32) dsum = xor_sum<32>(dsum);
                if (top <= 16) dsum = xor_sum<16>(dsum);
                if (top <= 8) dsum = xor_sum<8, true>(dsum);
                if (top <= 4) dsum = xor_sum<4, true>(dsum);
                if (top <= 2) dsum = xor_sum<2>(dsum);
                const double M = butterfly<1, 64>(__builtin_fmax(__builtin_fabs(rows.T_cur), __builtin_fabs(rows.T_new)));
                const double eta = ((M * 0x1p-46 + __builtin_fabs(dsum) * 0x1p-51) * 1.0625 + 0x1p-49) * a.bound_scale;
                const double ex = exp_v8(dsum);
                const bool valid = lane64 < top && ((inb_s >> lane64) & 1ull) != 0ull;
                const bool sure_acc = ex * (1.0 - eta) > sw_u, sure_rej = ex * (1.0 + eta) < sw_u;
                const bool unsure = valid && !(eta < 0x1p-7 && (sure_acc || sure_rej));      // (a NaN or an infinity anywhere in the sums ends up here)
                if (__ballot(unsure) == 0ull) {
                  decided = true;
                  acc_mask = __ballot(valid && sure_acc);
                  if (acc_mask != 0ull) {
                    const bool mine = ((acc_mask >> (rows.comp & 63)) & 1ull) != 0ull;
                    lp_curr = butterfly<1, 64>(mine ? rows.T_new : rows.T_cur);
                    if (lane64 < top && ((acc_mask >> lane64) & 1ull) != 0ull) Sme[sb + lane64] = sw_prop;
                    if constexpr (TracksState<Model>::value) Model::sweep_commit_all(cache, rows, acc_mask, sw_prop, sub, a.d);
                  }
                }
              }
            }
            if (!decided)
            for (int t = 0; t < d_len; ++t) {
              const int c = __builtin_amdgcn_readlane(ord, t);
              if (!((inb_s >> c) & 1ull)) continue;
              const double prop_c = lane_value(sw_prop, c), u_c = lane_value(sw_u, c);
              const double Tv = rows.comp == c ? rows.T_new : T_cur;
              const double prop_lp = butterfly<1, 64>(Tv);
              if (accept_sweep(prop_lp - lp_curr, u_c)) { lp_curr = prop_lp; T_cur = Tv; set_state(sb + c, prop_c); acc_mask |= 1ull << c; }
            }
            if (lane64 < d_len) {
              const bool inb_l = ((inb_s >> lane64) & 1ull) != 0ull, acc_l = ((acc_mask >> lane64) & 1ull) != 0ull;
              if (inb_l) TOTme[sb + lane64] += 1u + (acc_l ? 0x10000u : 0u);
              if (adapt[sb + lane64] != 0) adapt_component(sb + lane64, acc_l, CNTme[sb + lane64], cc[sb + lane64].batch_size, live, true);
            }
            Model::sweep_done(cache, rows, rows.comp >= 0 && ((acc_mask >> (rows.comp & 63)) & 1ull) != 0ull);
            if constexpr (kCert) { lpA = lp_curr; epsA = 0.0; }
            // the walk moves past the parameter: its first component was handed out when the previous slot looked ahead
            e = 0; e_top = 0; e_in = 0; ++np;
            slot += d_len - 1;
            if (slot + 1 < P_stepped) nx = prefetch(next_comp());
            continue;
          }
          sw_left = d_len;      // drawn, but the sums could not be prepared: update by update, proposals and uniforms taken from the lanes
          }
#endif
        }
      }
      const SlotPre me = nx;
      const int comp = me.comp;
      // bounds and type: requested now, needed after the proposal is drawn (the adaptation constants only at a batch boundary)
      const double k_lower = cc[comp].lower, k_upper = cc[comp].upper;
      const int k_type = cc[comp].type;
      if (BinaryOf<Model>::value && k_type == kTypeBinary) {   // compiled in only for models that may have binary parameters
        // ---- BinaryStepper.step (mcmc.js:753-767): both states evaluated, 0 chosen with
        // probability exp(z - log(exp(z) + exp(o))) after subtracting the larger log density
        const double old = me.cur;
        set_state(comp, 0.0);
        const double zero_ld = log_post<Model, G, kPassU>(S, a, data_lds, sub, xw, cache);
        set_state(comp, 1.0);
        const double one_ld = log_post<Model, G, kPassU>(S, a, data_lds, sub, xw, cache);
        const double mx = js_max2(zero_ld, one_ld);
        const double z = zero_ld - mx, o = one_ld - mx;
        const double zero_prob = exp_v8(z - log_v8(exp_v8(z) + exp_v8(o)));
        double now = 1.0;
        lp_curr = one_ld;
        if (rng.next() < zero_prob) { set_state(comp, 0.0); now = 0.0; lp_curr = zero_ld; }
        if (counter) (void)__hip_atomic_fetch_add(&TOTme[comp], 1u + ((now != old) ? 0x10000u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // evaluations and flips
        if (slot + 1 < P_stepped) nx = prefetch(next_comp());
        continue;
      }
      AMWG_PHASE(14);
      // ---- OnedimMetropolisStepper.step (mcmc.js:517-553)
      const double cur = me.cur;
      const bool in_sweep = kSweep && sw_left > 0;      // (this update's proposal and uniform were drawn when the sweep began)
      double prop;
      if (in_sweep) prop = lane_value(sw_prop, comp - sb);
      else {
        prop = rnorm_js(rng, cur, me.sd);
        if (chain_true<G>(k_type == kTypeInt)) prop = js_round(prop);
      }
      const bool inb = chain_true<G>(!(prop < k_lower || prop > k_upper));
      // the accept test's uniform (mcmc.js:528) is the next one of the stream whatever log_post returns: drawn now
      double u_accept = 0.0;
      if (inb) { set_state(comp, prop); u_accept = in_sweep ? lane_value(sw_u, comp - sb) : rng.next(); }
      if (in_sweep) --sw_left;
      // everything this slot draws is drawn: the next slot's component is known (and, if a multidimensional parameter begins there,
      // shuffled), and what the stepper needs of it is requested NOW, under the evaluation below
      AMWG_PHASE(8);
      if (slot + 1 < P_stepped) nx = prefetch(next_comp());
      AMWG_PHASE(9);
      bool accepted = false;
      bool certified = false;
      [[maybe_unused]] double aud_E_prop = 0.0;
      if constexpr (kCert) {
        // (the model's pass is the WAVEFRONT's: with a lane per chain the 64 chains of a wave are evaluated together, every lane taking part whether its own
        // proposal needs a value or not -- control flow is uniform here: the slot loop is, and the lanes have come back together from their rnorm loops)
        if (__ballot(inb) != 0ull) {
          wave_priority(0);
          const typename Model::Approx r = Model::template log_post_approx<G, BT>(cache, S, a.mc, a.d, data_lds, sub);
          wave_priority(kStepperPriority);
          AMWG_PHASE(10);
          if (inb) {
            const double dA = r.value - lpA;
            const double eta = ((r.eps + epsA + __builtin_fabs(dA) * 0x1p-51) * 1.0625 + 0x1p-49) * a.bound_scale;
            const double ex = exp_v8(dA);
            if constexpr (kAudit) { if (a.audit_adversarial) u_accept = audit_adversarial_u(ex, eta, u_accept, step + slot); }
            const int verdict = certified_test(dA, eta, ex, u_accept);      // (0 for a NaN anywhere)
            if constexpr (kAudit) {      // BOUND AUDIT: the expression's value of the proposal (the state holds it), beside the cheap one
              aud_E_prop = expression();
              audit_value(r.value, r.eps, aud_E_prop);
              audit_difference(dA, (r.eps + epsA + __builtin_fabs(dA) * 0x1p-51) * 1.0625, aud_E_prop - audE, verdict, u_accept);
            }
            if (chain_true<G>(verdict > 0)) { certified = true; accepted = true; lpA = r.value; epsA = r.eps; lp_exact = false; }
            else if (chain_true<G>(verdict < 0)) { certified = true; set_state(comp, cur); }
            if (certified && counter) (void)__hip_atomic_fetch_add(&TOTme[comp], 1u + (accepted ? 0x10000u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
      if (certified) {
      } else if (inb) {
        double prop_lp = 0.0;
        if constexpr (kCert) {      // the cheap values could not decide: the expression, for the current state first if a cheap value has been standing in for it
          // (ONE call site for both: the reference-order evaluation is an out-of-line call, and every call site costs the loop saved registers)
          for (int which = chain_true<G>(lp_exact) ? 1 : 0; which < 2; ++which) {
            if (which == 0) set_state(comp, cur);
            const double v = expression();
            if (which == 0) { lp_curr = v; lp_exact = true; set_state(comp, prop); }
            else prop_lp = v;
          }
        } else {
          prop_lp = expression();
        }
#if defined(__HIP_DEVICE_COMPILE__)
        // the next slot's prefetched values are "used" HERE: the wait for them lands right behind the pass's own LDS reads (which returned
        // after them -- no stall), instead of at the top of the next slot behind this slot's closing stores and counter update
        asm volatile("" : : "v"(nx.cur), "v"(nx.sd), "v"(nx.batch_size), "v"(nx.cnt.x), "v"(nx.cnt.y), "v"((int)nx.adapting));
#endif
        // Math.exp(prop - curr) > Math.random() (mcmc.js:527-528).  For a difference >= 0 (incl. +inf) the exponential is >= 1 > u, below
        // -746 it is exactly 0 (never > u): the decision is the reference's without evaluating it; NaN takes the general path (false).
        const double diff = prop_lp - lp_curr;
        if (chain_true<G>(diff >= 0.0)) accepted = true;
        else if (chain_true<G>(diff < -746.0)) accepted = false;
        else {
          // For d < 0:  1 + d <= exp(d) <= 1 + d + d*d/2, and V8's exp is within one ulp (< 2^-53 here) of exp: a uniform below the lower bound
          // or above the upper one (each taken with a margin of 2^-50, an order of magnitude more than the roundings of the bounds themselves
          // plus that ulp) decides the comparison exactly as the exponential would -- which is then evaluated for the band in between only
          // (about a quarter of the proposals at a 44 % acceptance rate); the ~45 instructions of exp are the longest single dependent chain
          // of an update.  Chains sharing a wavefront diverge here; the exponential runs for those that need it.
          const double lower = 1.0 + diff;
          if (chain_true<G>(u_accept < lower - 0x1p-50)) accepted = true;
          else if (chain_true<G>(diff > -1.0 && u_accept > (lower + 0.5 * diff * diff) + 0x1p-50)) accepted = false;
          else accepted = chain_true<G>(exp_v8(diff) > u_accept);
        }
        accepted = chain_true<G>(accepted);
        if (accepted) lp_curr = prop_lp;
        else set_state(comp, cur);
        if constexpr (kCert) { lpA = lp_curr; epsA = 0.0; }
        if (counter) (void)__hip_atomic_fetch_add(&TOTme[comp], 1u + (accepted ? 0x10000u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // run totals (not in the reference; parity tests compare them with the oracle's)
      }
      if constexpr (kAudit) { if (inb && accepted) audE = aud_E_prop; }      // (BOUND AUDIT: E follows the state)
      AMWG_PHASE(11);
      if (chain_true<G>(me.adapting)) adapt_component(comp, accepted, me.cnt, me.batch_size, writer);
      AMWG_PHASE(12);
    }
  }

  // the register mirror of the state (kTracksState models) is a second copy that every store must keep current (set_state -> on_set): once per
  // launch it is compared with the LDS copy, bit for bit -- a store that bypassed set_state would otherwise go unnoticed until a parity test
  if constexpr (kCert) {      // what leaves the launch: the expression's value of the final state if the host asked for it, else the pair the stepper holds
    if (a.finalize_lp && !lp_exact) { lp_curr = expression(); lp_exact = true; }
    if (!lp_exact) lp_curr = lpA;
    if (writer) cold_args()->ch.lp_eps[cl] = lp_exact ? 0.0 : epsA;
  }
  if constexpr (!GL && MirrorCheckOf<Model>::value) {
    if (!Model::template mirror_ok<G>(cache, S, a.d, sub)) (void)atomicOr(cold_args()->ch.error, kErrMirrorOutOfSync);
  }
#if defined(AMWG_X_PHASES)
  if ((tid & 63) == 0 && live) {
    unsigned long long *const hh = cold_args()->ch.audit_hist + 64;
#pragma unroll
    for (int q = 0; q < 16; ++q) (void)atomicAdd(hh + q, (unsigned long long)ph_acc[q]);
  }
#endif
  if constexpr (kAudit) {
    if (writer) {
      double *const au = cold_args()->ch.audit;
      const double v0 = au[cl], d0 = au[C + cl];
      au[cl] = (aud_max_v > v0 || !(aud_max_v == aud_max_v)) ? aud_max_v : v0;
      au[C + cl] = (aud_max_d > d0 || !(aud_max_d == aud_max_d)) ? aud_max_d : d0;
      au[2 * C + cl] += (double)aud_n;
      au[3 * C + cl] += (double)aud_bad;
    }
  }
  if (writer) {
    const cold_args_ptr ca = cold_args();
    double *const o_state = ca->ch.state, *const o_pls = ca->ch.prop_log_scale;
    int32_t *const o_ac = ca->ch.acceptance_count, *const o_it = ca->ch.iterations_since_adaption, *const o_bc = ca->ch.batch_count;
    int32_t *const o_acc = ca->ch.accepts, *const o_inb = ca->ch.inbounds;
    for (int p = 0; p < P; ++p) {
      o_state[p * C + cl] = S(p);
      o_ac[p * C + cl] = CNTme[p].x;
      o_it[p * C + cl] = CNTme[p].y;
      o_acc[p * C + cl] += (int32_t)(TOTme[p] >> 16);
      o_inb[p * C + cl] += (int32_t)(TOTme[p] & 0xffffu);
      if constexpr (kMulti) { o_pls[p * C + cl] = LOGPLSme[p]; o_bc[p * C + cl] = BCme[p]; }
    }
    if (wide_perm) { uint16_t *const o_p16 = ca->ch.perm16; for (int k = 0; k < n_named; ++k) o_p16[(int64_t)k * C + cl] = (uint16_t)pcol.get(k); }
    else ca->ch.perm[cl] = perm;
    ca->ch.rng_n[cl] = rng.consumed();
    ca->ch.lp_curr[cl] = lp_curr;
  }
}

// BT = the workgroup size class the instantiation is compiled for (its register budget): 1024 threads leave 128 VGPRs per lane,
// 512 leave 256, 256 and fewer 512.  Round 2 compiled everything for 1024 and paid for it with scratch spills inside the slot loop
// even where the launch used 256- or 512-thread workgroups (cfg2, cfg4).
// (Model::kMinWavesPerSimd, optional: the occupancy the register allocation must leave room for -- HIP's second __launch_bounds__ argument
// counts waves per SIMD)
template <class M, class = void> struct MinWavesOf { static constexpr int value = 1; };
template <class M> struct MinWavesOf<M, void_of<decltype(M::kMinWavesPerSimd)>> { static constexpr int value = M::kMinWavesPerSimd; };

template <class Model, int G, int BT>
__global__ void __launch_bounds__(BT, MinWavesOf<Model>::value) amwg_step_kernel(const StepArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  step_body<Model, G, BT>(a, smem);
}
// the kernel of a family with the lane-local re-evaluation (Model::kLaneReuse) in its row layout, a chain on one whole wavefront: the ordinary stepper
// plus the sweep prefetch (step_body: kSweep)
template <class Model, int BT>
__global__ void __launch_bounds__(BT, MinWavesOf<Model>::value) amwg_sweep_kernel(const StepArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  step_body<Model, 64, BT, false, true>(a, smem);
}
// the kernels that DECIDE FROM CERTIFIED VALUES (step_body: kCert -- options.full_evaluation = 0 for a model and lane count that has them): the ordinary one
// (the Normal family at one lane per chain, the Poisson family at 16) and the sweep kernel (the hierarchical family)
template <class Model, int G, int BT>
__global__ void __launch_bounds__(BT, MinWavesOf<Model>::value) amwg_step_kernel_cert(const StepArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  step_body<Model, G, BT, false, false, true>(a, smem);
}
template <class Model, int BT>
__global__ void __launch_bounds__(BT, MinWavesOf<Model>::value) amwg_sweep_kernel_cert(const StepArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  step_body<Model, 64, BT, false, true, true>(a, smem);
}
// the group-local kernel of a family that has one (amwg_gl.h): a chain on one whole wavefront
template <class Model, int BT>
__global__ void __launch_bounds__(BT, MinWavesOf<Model>::value) amwg_gl_kernel(const StepArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  step_body<Model, 64, BT, true>(a, smem);
}

}  // namespace amwg
