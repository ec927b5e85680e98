// amwg_pass.h -- the hand-scheduled pass over the data of a normal likelihood with loop-invariant sd, shared by the built-in
// families (amwg_models.h) and by translated closures (amwg_user.h: `for (i...) lp += ld.norm(x[i], mean, sd)` compiles to it).
#pragma once
#include "amwg_div.h"
#include "amwg_types.h"

namespace amwg {

// ---------------------------------------------------------------------------------------------
// The pass over the data of a normal likelihood with loop-invariant sd -- sum_i [ c - (x_i - m_i)^2 / den ] added term by term
// to `acc` in increasing i -- software-pipelined by hand.  Per observation it is the same eight fp64 operations, with the same
// roundings in the same order, as NormalModel::term<true> / div_by_invariant (amwg_div.h):
//     t = x - m;  tt = t*t;  q0 = tt*y.lo;  q1 = fma(tt, y.hi, q0);  r = fma(-den, q1, tt);  q = fma(r, y.hi, q1);  term = c - q;  acc += term
// and the terms are added in the same order, so the sum is bit-identical to the plain loop (pass_over_data).  What changes is the
// SCHEDULE: the compiler's own schedule of the plain loop waits for every LDS read right after issuing it and works through the
// terms two at a time, i.e. one long dependent chain per wave (rocprofv3, round 1: 0.74 of the fp64 issue rate at 4 waves per
// SIMD, 39 % of wave cycles waiting).  Here a block of U observations moves through the eight steps as eight STAGES of U
// independent instructions each (sched_barrier between stages keeps the compiler from re-serialising them), the LDS reads of
// block k+1 (and, for a gathered mean, the index reads of block k+2) are issued before the arithmetic of block k starts, and the
// U dependent `acc +=` of block k-1 are spread one per stage over block k -- no instruction waits for the one before it.
// Mean = a functor: m(i) for a per-observation mean gathered from LDS (HierNormalModel), or a constant (NormalModel).
#if defined(__HIP_DEVICE_COMPILE__)
#define AMWG_STAGE_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define AMWG_STAGE_FENCE() ((void)0)
#endif

template <int U>
struct NormBlock { double v[U]; };     // x, then t, tt, r in place

// "the loads of THIS block have landed": with a constant mean a block is four LDS reads (ds_read2_b64 / ds_read_b128 pairs of the
// U = 8 observations) and the four reads of the NEXT block were issued just before, so lgkmcnt <= 4 is exactly that.  Saying it
// once keeps the compiler from placing one s_waitcnt in front of every pair of subtractions (with one wave per SIMD -- one lane
// per chain at cfg2 -- every s_waitcnt is an issue slot the fp64 pipe idles through).  Only a hint: the compiler's own counter
// tracking still inserts whatever wait a different instruction selection would need.
#if defined(__HIP_DEVICE_COMPILE__)
#define AMWG_WAIT_LDS_LE(n) __builtin_amdgcn_s_waitcnt(0xC07F | ((n) << 8))   // vmcnt / expcnt untouched
#else
#define AMWG_WAIT_LDS_LE(n) ((void)0)
#endif

template <int U, bool ADD_PREV>
AMWG_HD void norm_block_stages(NormBlock<U> &b, const double (&m)[U], NormBlock<U> &q, const NormBlock<U> &prev, double &acc,
                                                  double c, double den, Reciprocal y) {
  // prev = the finished terms of the block before this one; its U additions are dealt over the seven stages (U = 8: one per
  // stage, two in the last)
  int a = 0;
  auto add_prev = [&](int upto) {
    if constexpr (ADD_PREV) { for (; a < upto && a < U; ++a) acc = acc + prev.v[a]; }
  };
#pragma unroll
  for (int u = 0; u < U; ++u) b.v[u] = b.v[u] - m[u];
  add_prev(1 * U / 7);
  AMWG_STAGE_FENCE();
#pragma unroll
  for (int u = 0; u < U; ++u) b.v[u] = b.v[u] * b.v[u];
  add_prev(2 * U / 7);
  AMWG_STAGE_FENCE();
#pragma unroll
  for (int u = 0; u < U; ++u) q.v[u] = b.v[u] * y.lo;
  add_prev(3 * U / 7);
  AMWG_STAGE_FENCE();
#pragma unroll
  for (int u = 0; u < U; ++u) q.v[u] = __builtin_fma(b.v[u], y.hi, q.v[u]);
  add_prev(4 * U / 7);
  AMWG_STAGE_FENCE();
#pragma unroll
  for (int u = 0; u < U; ++u) b.v[u] = __builtin_fma(-den, q.v[u], b.v[u]);
  add_prev(5 * U / 7);
  AMWG_STAGE_FENCE();
#pragma unroll
  for (int u = 0; u < U; ++u) q.v[u] = __builtin_fma(b.v[u], y.hi, q.v[u]);
  add_prev(6 * U / 7);
  AMWG_STAGE_FENCE();
#pragma unroll
  for (int u = 0; u < U; ++u) q.v[u] = c - q.v[u];
  add_prev(U);
  AMWG_STAGE_FENCE();
}

// x: LDS (or global) array of the observations; lane `sub` of the chain's G lanes takes observations sub, sub + G, ...
// MeanOf::gather == false: constant mean;  true: index array g (u8) and the chain's state S, mean of observation i = S(g[i])
// XT: storage type of the observations (double, or the u8 / i32 the translator picks for all-integer data arrays: exact conversions)
template <int G, int U, bool GATHER, class XT = double>
AMWG_HD double norm_pass_staged(const XT *x, const uint8_t *g, const StateView S, double mean, double c, double den,
                                                   Reciprocal y, int n_obs, int sub, double acc) {
  const int n_full = n_obs / G, rem = n_obs % G;
  const int n_blocks = n_full / U;
  int k = 0;
  if (n_blocks > 0) {
    NormBlock<U> xa, xb, qa, qb;
    double ma[U], mb[U];
    int ga[U], gb[U];     // GATHER: group indices, read one block further ahead than the values
    const XT *px = x + sub;
    const uint8_t *pg = g + sub;
    // (index prefetches past the end re-read the last block: harmless, and the loop body stays ONE basic block -- a conditional
    // load splits it, the stage fences stop holding across the pieces and the register sets get copied instead of swapped)
    auto load_idx = [&](int blk, int (&gi)[U]) {
      if constexpr (GATHER) {
        blk = blk < n_blocks ? blk : n_blocks - 1;
#pragma unroll
        for (int u = 0; u < U; ++u) gi[u] = pg[(blk * U + u) * G];
      }
    };
    auto load_val = [&](int blk, NormBlock<U> &xv, double (&mv)[U], const int (&gi)[U]) {
#pragma unroll
      for (int u = 0; u < U; ++u) xv.v[u] = (double)px[(blk * U + u) * G];
#pragma unroll
      for (int u = 0; u < U; ++u) mv[u] = GATHER ? S(gi[u]) : mean;
    };
    // prologue: block 0 loaded, indices of block 1 on their way
    load_idx(0, ga);
    load_val(0, xa, ma, ga);
    load_idx(1, gb);
    AMWG_STAGE_FENCE();
    // block 0: nothing to add yet
    if (n_blocks > 1) { load_val(1, xb, mb, gb); load_idx(2, ga); }
    AMWG_STAGE_FENCE();
    norm_block_stages<U, false>(xa, ma, qa, qa, acc, c, den, y);
    int blk = 1;
    // steady state, two blocks per trip (A/B register sets swap roles, no copies): on entry the terms of block blk-1 sit in qa,
    // the values of block blk in xb/mb, the indices of block blk+1 in ga
    for (; blk + 2 < n_blocks; blk += 2) {
      load_val(blk + 1, xa, ma, ga);
      load_idx(blk + 2, gb);
      AMWG_STAGE_FENCE();
      if constexpr (!GATHER && U == 8 && sizeof(XT) == 8) { AMWG_WAIT_LDS_LE(4); AMWG_STAGE_FENCE(); }
      norm_block_stages<U, true>(xb, mb, qb, qa, acc, c, den, y);
      load_val(blk + 2, xb, mb, gb);
      load_idx(blk + 3, ga);
      AMWG_STAGE_FENCE();
      if constexpr (!GATHER && U == 8 && sizeof(XT) == 8) { AMWG_WAIT_LDS_LE(4); AMWG_STAGE_FENCE(); }
      norm_block_stages<U, true>(xa, ma, qa, qb, acc, c, den, y);
    }
    // epilogue: one or two blocks left (blk, and maybe blk + 1), terms of blk-1 pending in qa
    if (blk < n_blocks) {
      if (blk + 1 < n_blocks) load_val(blk + 1, xa, ma, ga);
      AMWG_STAGE_FENCE();
      norm_block_stages<U, true>(xb, mb, qb, qa, acc, c, den, y);
      if (blk + 1 < n_blocks) {
        norm_block_stages<U, true>(xa, ma, qa, qb, acc, c, den, y);
#pragma unroll
        for (int u = 0; u < U; ++u) acc = acc + qa.v[u];
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) acc = acc + qb.v[u];
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) acc = acc + qa.v[u];
    }
    k = n_blocks * U;
  }
  // ragged tail: fewer than U whole rounds are left, and the lanes below n_obs % G take one more observation.  ONE masked block instead
  // of a loop of dependent single terms (with a handful of chains and short data loops that loop was a fifth of an update): every lane
  // runs the eight stages on U slots, slots beyond its own count re-read its last valid observation, and only the first cnt terms are
  // added -- in the same order as before (whole rounds first, the remainder observation last).
  const int left = n_full - k;                              // 0 .. U-1, uniform
  if (left + (rem > 0 ? 1 : 0) <= U / 2 + 1) {
    // a few terms only (e.g. 10^4 observations on 64 lanes: 19 blocks, then 4 rounds and a quarter): term by term -- (left + 1) x 8
    // instructions instead of the 8 x U + selects of the masked block below.  The chain of dependent operations this leaves is latency the
    // SIMD's other wave fills; with few waves per SIMD and longer tails the masked block is the better trade.
    // every value of the tail is requested before the first term is computed: one LDS round trip for the tail, not one per round
    constexpr int TMAX = U / 2 + 1;
    double xv[TMAX], mv[TMAX];
    const int cnt_all = left + (rem > 0 ? 1 : 0);            // uniform: rounds in the tail, the remainder round (if any) last
    if (cnt_all > 0) {
#pragma unroll
    for (int r = 0; r < TMAX; ++r) {
      int i = (k + (r < cnt_all ? r : 0)) * G + sub;
      i = i < n_obs ? i : n_obs - 1;                         // (the remainder round of a lane without an observation in it: any valid address)
      xv[r] = (double)x[i];
      mv[r] = GATHER ? S(g[i]) : mean;
    }
    AMWG_STAGE_FENCE();
#pragma unroll
    for (int r = 0; r < TMAX; ++r) {
      if (r < cnt_all) {                                     // uniform
        const double t = xv[r] - mv[r];
        const double term = c - div_by_invariant(t * t, den, y);
        const bool mine = r < left || sub < rem;             // the remainder round belongs to the lanes below rem only
        acc = mine ? acc + term : acc;
      }
    }
    }
  } else if (left > 0 || rem > 0) {
    const int cnt = left + (sub < rem ? 1 : 0);             // this lane's terms: 0 .. U
    NormBlock<U> xt, qt;
    double mt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int r = u < cnt ? u : (cnt > 0 ? cnt - 1 : 0);
      int i = (k + r) * G + sub;
      i = i < n_obs ? i : n_obs - 1;                        // (a lane without any term: any valid address)
      xt.v[u] = (double)x[i];
      mt[u] = GATHER ? S(g[i]) : mean;
    }
    AMWG_STAGE_FENCE();
    norm_block_stages<U, false>(xt, mt, qt, qt, acc, c, den, y);
#pragma unroll
    for (int u = 0; u < U; ++u) acc = u < cnt ? acc + qt.v[u] : acc;
  }
  return acc;
}

// ONE lane per chain (the reference's own summation order): all 64 lanes of a wave -- 64 different chains -- need the SAME
// observation at the same time, so the observations are wave-uniform.  They are then read through the SCALAR path (s_load_dwordx16 =
// eight observations per instruction, from global memory through the scalar cache / L2, no LDS tile at all) and enter the fp64
// pipe as SGPR operands.  With one wave per SIMD (65 536 chains = 1024 waves on 1024 SIMDs) every instruction that is not an fp64
// operation is an issue slot the pipe idles through: the LDS version spends 8 ds_read + 9 others per 128 fp64 operations, this
// one 2 s_load + ~8.  Scalar loads return out of order, so the only wait is lgkmcnt(0): a chunk of 16 observations is requested
// one whole chunk (~520 cycles of arithmetic) before it is needed.  Same operations, same order, same bits as the plain loop.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const double __attribute__((address_space(4))) *amwg_uniform_f64_ptr;
#else
typedef const double *amwg_uniform_f64_ptr;
#endif

struct StagedFirst { static constexpr bool value = true; };
struct StagedLater { static constexpr bool value = false; };

template <int U>
AMWG_HD double norm_pass_uniform(const double *x_global, double mean, double c, double den, Reciprocal y, int n_obs, double acc) {
  constexpr int CH = 2 * U;                       // observations per chunk (two blocks of U)
  const int n_chunks = n_obs / CH;
  int k = 0;
  if (n_chunks > 0) {
    amwg_uniform_f64_ptr px = (amwg_uniform_f64_ptr)(uintptr_t)x_global;
    double ma[U];
#pragma unroll
    for (int u = 0; u < U; ++u) ma[u] = mean;
    NormBlock<CH> sa, sb;                         // the two chunks in flight (wave-uniform: scalar registers)
    NormBlock<U> ta, tb, qa, qb;
    auto load_chunk = [&](int ch, NormBlock<CH> &dst) {
      ch = ch < n_chunks ? ch : n_chunks - 1;     // prefetches past the end re-read the last chunk (keeps the loop one basic block)
#pragma unroll
      for (int u = 0; u < CH; ++u) dst.v[u] = px[ch * CH + u];
    };
    auto lo = [&](const NormBlock<CH> &src, NormBlock<U> &t) {
#pragma unroll
      for (int u = 0; u < U; ++u) t.v[u] = src.v[u];
    };
    auto hi = [&](const NormBlock<CH> &src, NormBlock<U> &t) {
#pragma unroll
      for (int u = 0; u < U; ++u) t.v[u] = src.v[U + u];
    };
    // A scalar wait drains EVERY outstanding scalar load, so a new request must be issued right AFTER the wait for the previous
    // one, never before it: per chunk  { wait (the chunk requested one chunk ago has landed) ; request the next chunk into the
    // other register set ; 2 x 7 stages on this chunk }.
    auto compute = [&](const NormBlock<CH> &src, auto first) {
      lo(src, ta);
      norm_block_stages<U, !decltype(first)::value>(ta, ma, qa, qb, acc, c, den, y);
      hi(src, tb);
      norm_block_stages<U, true>(tb, ma, qb, qa, acc, c, den, y);
    };
    load_chunk(0, sb);
    AMWG_WAIT_LDS_LE(0); AMWG_STAGE_FENCE();
    load_chunk(1, sa);
    AMWG_STAGE_FENCE();
    compute(sb, StagedFirst{});                           // chunk 0: its first block has no earlier terms to add
    int ch = 1;
    for (; ch + 1 < n_chunks; ch += 2) {          // chunk ch is on its way into sa
      AMWG_WAIT_LDS_LE(0); AMWG_STAGE_FENCE();
      load_chunk(ch + 1, sb);
      AMWG_STAGE_FENCE();
      compute(sa, StagedLater{});
      AMWG_WAIT_LDS_LE(0); AMWG_STAGE_FENCE();
      load_chunk(ch + 2, sa);
      AMWG_STAGE_FENCE();
      compute(sb, StagedLater{});
    }
    if (ch < n_chunks) {                          // one chunk left, on its way into sa
      AMWG_WAIT_LDS_LE(0); AMWG_STAGE_FENCE();
      compute(sa, StagedLater{});
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc = acc + qb.v[u];
    k = n_chunks * CH;
  }
  for (; k < n_obs; ++k) {                        // ragged tail
    const double t = x_global[k] - mean;
    acc += c - div_by_invariant(t * t, den, y);
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------
// The CERTIFIED pass, one lane per chain: S2 = sum_i (x_i - mean)^2 over all observations -- two fp64 operations per observation (sub, fma)
// instead of the eight of the reference's term c - (x - mean)^2 / den.  As real numbers  sum_i [c - (x_i - mean)^2 / den] = n c - S2 / den:
// the stepper's accept test needs log_post only to within what decides exp(difference) > u, and both this sum and the term-by-term one are
// within a computable bound of that real number (amwg_models.h NormalModel::log_post_approx; amwg_kernel.h "certified decisions").  The order
// of the additions is free here (eight interleaved partial sums: no dependent chain), the observations arrive through the scalar path as in
// norm_pass_uniform.
template <int U>
AMWG_HD double norm_sq_pass_uniform(const double *x_global, double mean, int n_obs) {
  constexpr int CH = 2 * U;
  double acc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) acc[u] = 0.0;
  const int n_chunks = n_obs / CH;
  int k = 0;
  if (n_chunks > 0) {
    amwg_uniform_f64_ptr px = (amwg_uniform_f64_ptr)(uintptr_t)x_global;
    NormBlock<CH> sa, sb;                         // the two chunks in flight (wave-uniform: scalar registers)
    auto load_chunk = [&](int ch, NormBlock<CH> &dst) {
      ch = ch < n_chunks ? ch : n_chunks - 1;     // prefetches past the end re-read the last chunk (keeps the loop one basic block)
#pragma unroll
      for (int u = 0; u < CH; ++u) dst.v[u] = px[ch * CH + u];
    };
    auto compute = [&](const NormBlock<CH> &src) {
#pragma unroll
      for (int u = 0; u < U; ++u) { const double t = src.v[u] - mean; acc[u] = __builtin_fma(t, t, acc[u]); }
      AMWG_STAGE_FENCE();
#pragma unroll
      for (int u = 0; u < U; ++u) { const double t = src.v[U + u] - mean; acc[u] = __builtin_fma(t, t, acc[u]); }
      AMWG_STAGE_FENCE();
    };
    load_chunk(0, sb);
    AMWG_WAIT_LDS_LE(0); AMWG_STAGE_FENCE();
    load_chunk(1, sa);
    AMWG_STAGE_FENCE();
    compute(sb);
    int ch = 1;
    for (; ch + 1 < n_chunks; ch += 2) {          // chunk ch is on its way into sa
      AMWG_WAIT_LDS_LE(0); AMWG_STAGE_FENCE();
      load_chunk(ch + 1, sb);
      AMWG_STAGE_FENCE();
      compute(sa);
      AMWG_WAIT_LDS_LE(0); AMWG_STAGE_FENCE();
      load_chunk(ch + 2, sa);
      AMWG_STAGE_FENCE();
      compute(sb);
    }
    if (ch < n_chunks) {                          // one chunk left, on its way into sa
      AMWG_WAIT_LDS_LE(0); AMWG_STAGE_FENCE();
      compute(sa);
    }
    k = n_chunks * CH;
  }
  for (; k < n_obs; ++k) {                        // ragged tail
    const double t = x_global[k] - mean;
    acc[0] = __builtin_fma(t, t, acc[0]);
  }
  double s = 0.0;
#pragma unroll
  for (int u = 0; u < U; ++u) s += acc[u];
  return s;
}

// (xor_partner is amwg_kernel.h's: the step kernel's header is included before this one wherever the device code below is compiled)
// ---------------------------------------------------------------------------------------------
// The certified pass of the Normal family, one lane per chain, for the 64 chains of a wavefront AT ONCE:  S2_c = sum_i (x_i - mu_c)^2.
// Lane l enters with its chain's mean and leaves with its chain's sum.  With a lane per chain every chain needs every observation: read one at a time
// and broadcast (scalar loads, norm_sq_pass_uniform) the pass waits for its loads -- two operations per observation are not enough work to cover a scalar
// cache round trip with one wavefront per SIMD (measured: 0.37 of the issue rate).  Here the ROLES are swapped for the length of the pass: a lane holds
// OBSERVATIONS (lane l: x_l, x_(l+64), ...: coalesced reads of the LDS tile, each value used for all 64 chains), the means are broadcast (v_readlane, eight
// chains' worth at a time), every lane keeps 64 partial sums -- one per chain -- and a transposing butterfly (v_permlane32_swap / v_permlane16_swap /
// DPP: 63 exchange-and-add steps) leaves chain c's total in lane c.  Same count of fp64 operations, no load on the critical path.  The order of the
// additions is whatever this schedule gives: the value is used with its rounding bound only (NormalModel::log_post_approx).
#if defined(__HIPCC__) || defined(__HIPCC_RTC__)      // (device code only; the host builds of these headers -- tests/host -- never call it)
#ifndef AMWG_WAVE_BLOCK
#define AMWG_WAVE_BLOCK 16
#endif
constexpr int kWaveBlock = AMWG_WAVE_BLOCK;      // observations per lane and block of the wavefront's pass (tools/build_variant.sh -DAMWG_WAVE_BLOCK=8: round 5's)
template <int N> struct PassBlock { static constexpr int value = N; };
// MEANS THROUGH THE SCALAR MEMORY PATH (round 6, last day).  The 64 means of a wavefront's chains reach the blocks' arithmetic as scalar operands; v_readlane -- two
// per mean and block, 1 536 of a cfg2 pass's 23 800 vector-issue slots -- is itself a vector instruction.  With a scratch line per wavefront in device memory
// (`scr`, 64 doubles: DataRef::wave_scratch) every lane STORES its mean once per pass; when the stores have reached L2 (vmcnt 0) the wavefront drops what the scalar
// cache may hold of the line from the pass before (s_dcache_inv: the scalar cache is read-only and not kept coherent with vector stores) and the blocks fetch eight means
// at a time with s_load_dwordx16 -- the scalar unit's own issue slots --, one group ahead of its use: the first block's loads come from L2, the later ones from the
// scalar cache.  cfg2: 1.435e9 -> 1.505e9 updates/s (2.38 -> 2.27 vector instructions per observation-lane).  Loads that bypass the scalar cache (glc) measured 1.24e9:
// ~1 300 cycles each, more than a group's arithmetic covers.  scr == nullptr (AMWG_WAVE_SCRATCH=0, or a host that allocated none): v_readlane.
typedef int amwg_v16i __attribute__((ext_vector_type(16)));
// this wavefront's line of DataRef::wave_scratch (nullptr when the host gave none)
__device__ __forceinline__ double *wave_scratch_of(const DataRef &d) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (!d.wave_scratch) return nullptr;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));      // (the same for all its lanes: a scalar register)
  return d.wave_scratch + (size_t)wave * 64;
#else
  (void)d; return nullptr;
#endif
}
template <int B>
__device__ __forceinline__ double norm_sq_pass_wave(const double *x, double mu, int n_obs, double *scr = nullptr) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int lane = (int)(threadIdx.x & 63u);
  double a[64];
#pragma unroll
  for (int c = 0; c < 64; ++c) a[c] = 0.0;
  const int mu_lo = (int)(uint32_t)f64_bits(mu), mu_hi = (int)(uint32_t)(f64_bits(mu) >> 32);
  auto mean_of = [&](int c) { return bits_f64(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(mu_hi, c) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane(mu_lo, c)); };
  bool smem_done = false;
#if !defined(AMWG_X_NO_SMEM_MEANS)
  if (scr != nullptr) {      // (wave-uniform)
    scr[lane] = mu;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");      // (the stores have reached L2; this wavefront's line of the scalar cache, if an earlier pass left one, is dropped)
    auto fetch = [&](int g) { amwg_v16i r; asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(r) : "s"(scr), "s"(g * 8)); return r; };
    auto landed = [&](amwg_v16i &r) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r)); };
    auto mean_in = [&](const amwg_v16i &r, int j) { return bits_f64(((uint64_t)(uint32_t)r[2 * j + 1] << 32) | (uint64_t)(uint32_t)r[2 * j]); };
    auto block_s = [&](auto tag, int at) {
      constexpr int BB = decltype(tag)::value;
      double xv[BB];
      amwg_v16i cur = fetch(0);
#pragma unroll
      for (int b = 0; b < BB; ++b) xv[b] = x[at + b * 64 + lane];
#pragma unroll
      for (int g = 0; g < 64; g += 8) {
        landed(cur);
        amwg_v16i nxt = cur;
        if (g + 8 < 64) nxt = fetch(g + 8);
#pragma unroll
        for (int b = 0; b < BB; ++b) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { const double t = xv[b] - mean_in(cur, j); a[g + j] = __builtin_fma(t, t, a[g + j]); }
        }
        cur = nxt;
      }
    };
    int base = 0;
    for (; base + 64 * B <= n_obs; base += 64 * B) block_s(PassBlock<B>{}, base);
    if constexpr (B >= 32) { if (base + 64 * 16 <= n_obs) { block_s(PassBlock<16>{}, base); base += 64 * 16; } }
    if constexpr (B >= 16) { if (base + 64 * 8 <= n_obs) { block_s(PassBlock<8>{}, base); base += 64 * 8; } }
    if constexpr (B >= 8) { if (base + 64 * 4 <= n_obs) { block_s(PassBlock<4>{}, base); base += 64 * 4; } }
    if constexpr (B >= 4) { if (base + 64 * 2 <= n_obs) { block_s(PassBlock<2>{}, base); base += 64 * 2; } }
    for (; base + 64 <= n_obs; base += 64) block_s(PassBlock<1>{}, base);
    if (base < n_obs) {
      const int i = base + lane;
      const bool has = i < n_obs;
      const double xv = x[has ? i : base];
      amwg_v16i cur = fetch(0);
#pragma unroll
      for (int g = 0; g < 64; g += 8) {
        landed(cur);
        amwg_v16i nxt = cur;
        if (g + 8 < 64) nxt = fetch(g + 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const double t = has ? xv - mean_in(cur, j) : 0.0; a[g + j] = __builtin_fma(t, t, a[g + j]); }
        cur = nxt;
      }
    }
    smem_done = true;
  }
#endif
  if (!smem_done) {
  // a block of BB observations per lane against all 64 means: the means travel eight at a time into scalar registers (16 v_readlane per 8 BB subtract-fma pairs, so
  // the longer the block the smaller their share: 2 + 2 / BB vector instructions per observation and chain -- 2.25 at eight, 2.06 at 32), the eight running sums interleave
  auto block = [&](auto tag, int at) {
    constexpr int BB = decltype(tag)::value;
    double xv[BB];
#pragma unroll
    for (int b = 0; b < BB; ++b) xv[b] = x[at + b * 64 + lane];
#pragma unroll
    for (int g = 0; g < 64; g += 8) {
      double m[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) m[j] = mean_of(g + j);
#pragma unroll
      for (int b = 0; b < BB; ++b) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const double t = xv[b] - m[j]; a[g + j] = __builtin_fma(t, t, a[g + j]); }
      }
    }
  };
  int base = 0;
  for (; base + 64 * B <= n_obs; base += 64 * B) block(PassBlock<B>{}, base);
  // the rest in blocks of half the length each (round 5 walked it 64 observations at a time: a quarter of a block's arithmetic for all of its broadcasts), the last round masked
  if constexpr (B >= 32) { if (base + 64 * 16 <= n_obs) { block(PassBlock<16>{}, base); base += 64 * 16; } }
  if constexpr (B >= 16) { if (base + 64 * 8 <= n_obs) { block(PassBlock<8>{}, base); base += 64 * 8; } }
  if constexpr (B >= 8) { if (base + 64 * 4 <= n_obs) { block(PassBlock<4>{}, base); base += 64 * 4; } }
  if constexpr (B >= 4) { if (base + 64 * 2 <= n_obs) { block(PassBlock<2>{}, base); base += 64 * 2; } }
  if constexpr (B >= 2) { if (base + 64 <= n_obs) { block(PassBlock<1>{}, base); base += 64; } }
  for (; base + 64 <= n_obs; base += 64) block(PassBlock<1>{}, base);
  if (base < n_obs) {
    const int i = base + lane;
    const bool has = i < n_obs;
    const double xv = x[has ? i : base];
#pragma unroll
    for (int g = 0; g < 64; g += 8) {
      double m[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) m[j] = mean_of(g + j);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const double t = has ? xv - m[j] : 0.0; a[g + j] = __builtin_fma(t, t, a[g + j]); }
    }
  }
  }      // (!smem_done)
  // transposing butterfly: after the step with offset o a lane holds the chains that agree with it in that bit, a[j] <- kept + received
  auto swap_add = [&](double A, double Bv, int off) {      // A: what the lanes with the bit CLEAR keep, Bv: what the lanes with the bit SET keep
    const uint32_t al = (uint32_t)f64_bits(A), ah = (uint32_t)(f64_bits(A) >> 32), bl = (uint32_t)f64_bits(Bv), bh = (uint32_t)(f64_bits(Bv) >> 32);
    if (off == 32) {
      const auto l = __builtin_amdgcn_permlane32_swap(al, bl, false, false), h = __builtin_amdgcn_permlane32_swap(ah, bh, false, false);
      return bits_f64(((uint64_t)h[0] << 32) | (uint64_t)l[0]) + bits_f64(((uint64_t)h[1] << 32) | (uint64_t)l[1]);
    } else {
      const auto l = __builtin_amdgcn_permlane16_swap(al, bl, false, false), h = __builtin_amdgcn_permlane16_swap(ah, bh, false, false);
      return bits_f64(((uint64_t)h[0] << 32) | (uint64_t)l[0]) + bits_f64(((uint64_t)h[1] << 32) | (uint64_t)l[1]);
    }
  };
#pragma unroll
  for (int j = 0; j < 32; ++j) a[j] = swap_add(a[j], a[j + 32], 32);      // the upper 32 lanes' a[j] <-> the lower 32 lanes' a[j + 32]: every lane then adds its two registers
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = swap_add(a[j], a[j + 16], 16);      // likewise between the odd and the even rows of 16 lanes
  {
    const bool up = (lane & 8) != 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const double send = up ? a[j] : a[j + 8], keep = up ? a[j + 8] : a[j]; a[j] = keep + xor_partner<8, true>(send); }
  }
  {
    const bool up = (lane & 4) != 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const double send = up ? a[j] : a[j + 4], keep = up ? a[j + 4] : a[j]; a[j] = keep + xor_partner<4, true>(send); }
  }
  {
    const bool up = (lane & 2) != 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) { const double send = up ? a[j] : a[j + 2], keep = up ? a[j + 2] : a[j]; a[j] = keep + xor_partner<2>(send); }
  }
  {
    const bool up = (lane & 1) != 0;
    const double send = up ? a[0] : a[1], keep = up ? a[1] : a[0];
    a[0] = keep + xor_partner<1>(send);
  }
  return a[0];
#else
  (void)x; (void)mu; (void)n_obs;
  return 0.0;
#endif
}
#endif

}  // namespace amwg
