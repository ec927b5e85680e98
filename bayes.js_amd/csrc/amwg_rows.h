// amwg_rows.h -- lane-local re-evaluation and sweep prefetch for TRANSLATED closures (a chain on one whole wavefront).
//
// The reference treats every log_post alike (mcmc.js:524-526, 685-688): the whole closure is evaluated for every update.  For a closure that
// ENDS in the likelihood loop of a model with group means,
//     for (i = 0; i < y.length; i++) lp += ld.norm(y[i], state.theta[g[i]], sd);   return lp;
// whose labels repeat with the lane stride (g[i] == g[i % 64]: a lane meets ONE group), the value a lane contributes to the 64-lane sum is a pure
// function of three numbers: the value its accumulator holds when that loop begins (`head`: everything the closure adds before -- priors, other
// loops --, computed by the generated code exactly as always), the mean of its observations (its one group's theta) and sd.  The same three
// numbers give the same bits, so a lane whose numbers did not change since it last formed its sum need not form it again: this is what the
// hand-written hierarchical family does (amwg_models.h lane_sum_rows / prefetch_rows), here for whatever head the closure has.  The generated
// model (bayes.js_amd/translate.js: "row plan") supplies
//     kRowN, kRowBase, kRowGroups, kRowY, kRowLabels, kRowDataMid, kRowSweep
//     head<64>(S, d, smem, sub)     this lane's accumulator at the loop (the closure's statements before it, lane-split as always)
//     row_sd(S, d)                  the loop's sd expression (reads of the state only)
// and inherits the rest from UserRows<UserModel>.  Results are IDENTICAL to evaluating everything (options.full_evaluation = 1 switches the row
// layout off; tests compare the two chain by chain and both with the reference's goldens): like the cached log_post of the current state this is
// work not done twice, not an approximation.
//
// Row layout in LDS (DataRef::pad = row pitch Rp, odd): tile [64][Rp] of y (row j = the observations of lane j), the first 64 labels, per
// wavefront kMaxLocal rows of terms, one 256-uniform window per wavefront (the sweep kernel's stream).  Other data arrays the head reads stay in
// global memory (they are read by lane 0 or by a short lane-split loop, once per evaluation).
#pragma once
#include "amwg_user.h"      // (which includes this file at its end: NormInv's formulas, user_arr)
#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
#include "amwg_kernel.h"    // butterfly (the certified values below); every device build includes it anyway
#endif
#if defined(__HIPCC__) || defined(__HIPCC_RTC__)      // (device code throughout: the host build of a generated model -- tests/host -- sees nothing of it)
#include "amwg_div.h"
#include "amwg_ld.h"
#include "amwg_math.h"
#include "amwg_pass.h"
#include "amwg_types.h"
#include "amwg_window.h"

namespace amwg {

struct RowsCache {
  double sd, c, den;      // the invariants of the row loop's sd, kept while sd does not change (NormInv of amwg_user.h, by value)
  Reciprocal y;
  bool inv_fast;
  // the last two sums this lane formed, each with what it was formed FROM
  double a_start, a_mean, a_sd, a_T, b_start, b_mean, b_sd, b_T;
  bool a_recent, loaded;
  int my_group;           // the label of this lane's observations (-1: the lane has none)
  // certified decisions (round 6, below): this lane's sum of squares S2 = sum (y_i - mean)^2 over its row, with the mean it was formed for
  double s2, s2_mean;
};

struct UserSweepRows { bool ok; double T_cur, T_new; int comp; bool new_in_b; };
// what a lane's head contributes, with the magnitudes of its additions and their number (generated: UserModel::head_pair; the certified values below)
struct HeadPair { double value, mag, cnt; };

template <class M>
struct UserRows {
  static constexpr bool kLaneReuse = true;
  static constexpr bool kDynamicLds = true;
  static constexpr int kMaxLocal = 4;      // lanes whose sums are re-formed cooperatively; more stale lanes: the ordinary pass over the rows
  using Cache = RowsCache;
  using SweepStream = WindowStream;
  __host__ __device__ static int row_pitch(int n_obs) { return ((n_obs + 63) / 64) | 1; }
  __host__ __device__ static int local_rows(int groups) { const int per_group = groups > 0 ? 64 / groups : 1; const int r = per_group < 2 ? 2 : per_group; return r > kMaxLocal ? kMaxLocal : r; }
  __host__ __device__ static int term_pitch(int pitch) { return (pitch + 16 + 1) & ~1; }
  __host__ __device__ static size_t rows_window_offset(int pitch, int waves, int groups) { return (size_t)64 * pitch * 8 + 64 + (size_t)waves * local_rows(groups) * term_pitch(pitch) * 8; }
  __host__ __device__ static size_t rows_lds_bytes(int pitch, int waves, int groups) { return rows_window_offset(pitch, waves, groups) + (size_t)waves * 256 * 8; }
  __host__ __device__ static size_t lds_bytes_of(const DataRef &d, int lanes, int threads) { return d.pad > 0 ? rows_lds_bytes(d.pad, threads / 64, M::kRowGroups) : M::lds_bytes(0, 0, lanes); }
  // which parameter vector a sweep prefetch is for (amwg_kernel.h kSweep): theta's place in the state
  __device__ __forceinline__ static size_t window_offset(const DataRef &d, int waves) { return rows_window_offset(d.pad, waves, M::kRowGroups); }
  __device__ __forceinline__ static int sweep_base(const DataRef &) { return M::kRowBase; }
  __device__ __forceinline__ static int sweep_len(const DataRef &) { return M::kRowSweep ? M::kRowGroups : -1; }
  __device__ __forceinline__ static const uint8_t *labels(const unsigned char *smem, int pitch) { return smem + (size_t)64 * pitch * 8; }
  // the one component (entry of theta) lane `sub` stands for in a sweep: the one whose term of the head it holds, else its group's
  __device__ __forceinline__ static int sweep_comp(const unsigned char *smem, const DataRef &d, int sub) {
    return sub < M::kRowGroups ? sub : (sub < M::kRowN ? (int)labels(smem, d.pad)[sub] : -1);
  }
  __device__ static void stage_rows(unsigned char *smem, const DataRef &d, int tid, int nt) {
    double *dst = reinterpret_cast<double *>(smem);
    const double *y = static_cast<const double *>(user_arr<M::kRowY>(d));
    const uint8_t *g = static_cast<const uint8_t *>(user_arr<M::kRowLabels>(d));
    const int Rp = d.pad;
    for (int i = tid; i < M::kRowN; i += nt) dst[(i & 63) * Rp + (i >> 6)] = y[i];
    uint8_t *gd = smem + (size_t)64 * Rp * 8;
    for (int i = tid; i < 64; i += nt) gd[i] = i < M::kRowN ? g[i] : 0;
  }
  __device__ __forceinline__ static Cache cache_init() {
    const double nan = __builtin_nan("");
    return Cache{nan, 0.0, 0.0, Reciprocal{0.0, 0.0}, false, nan, nan, nan, 0.0, nan, nan, nan, 0.0, false, false, -1, 0.0, nan};
  }
  __device__ __forceinline__ static void load(Cache &k, const unsigned char *smem, int pitch, int sub) {
    if (k.loaded) return;
    k.loaded = true;
    k.my_group = sub < M::kRowN ? (int)labels(smem, pitch)[sub] : -1;
  }
  // c, den and 1/den of the loop's sd: the same roundings in the same order as norm_inv (amwg_user.h), formed again only when sd changes
  // (the logarithm and the division live out of line -- values by register, two calls as for the hand-written family's norm_cache_cold_a / _b --: sd changes in one
  // update of a step's many, and inlined at every evaluation site the code was a tenth of the certified kernel's hot path)
  struct InvA { double c, den; };
  __device__ __attribute__((noinline)) static InvA inv_cold_a(double sd) { return InvA{norm_c(-0.5 * log_v8(2 * kPi), sd), norm_den(sd)}; }
  __device__ __attribute__((noinline)) static Reciprocal inv_cold_b(double den) { return make_reciprocal(den); }
  __device__ __forceinline__ static void update_inv(Cache &k, double sd) {
    if (f64_bits(sd) == f64_bits(k.sd)) return;
    const InvA a = inv_cold_a(sd);
    k.sd = sd;
    k.c = a.c;
    k.den = a.den;
    k.y = inv_cold_b(a.den);
    k.inv_fast = mid_range(a.den);
  }
  __device__ __forceinline__ static double lane_double(double v, int src) {      // v of lane `src` (wave-uniform)
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(f64_bits(v) >> 32), src);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)f64_bits(v), src);
    return bits_f64(((uint64_t)hi << 32) | (uint64_t)lo);
  }
  // the ordinary pass of a lane over ITS row: start + term_0 + term_1 + ... in order -- what norm_data_loop_gather<64, periodic> (amwg_user.h) computes
  template <int U>
  __device__ __forceinline__ static double rows_full(const Cache &k, bool fast, double mean, const double *row, int sub, double acc) {
    constexpr int n_full = M::kRowN >> 6, rem = M::kRowN & 63;
    const double last = row[sub < rem ? n_full : 0];      // (the remainder round's observation, requested before the pass)
    if (fast) {
      acc = norm_pass_staged<1, U, false>(row, nullptr, StateView{nullptr}, mean, k.c, k.den, k.y, n_full, 0, acc);
      const double t = last - mean;
      const double term = k.c - div_by_invariant(t * t, k.den, k.y);
      return sub < rem ? acc + term : acc;
    }
    for (int r = 0; r < n_full; ++r) { const double t = row[r] - mean; acc += k.c - (t * t) / k.den; }
    const double t = last - mean;
    const double term = k.c - (t * t) / k.den;
    return sub < rem ? acc + term : acc;
  }
  // may the passes take the 4-operation quotient of amwg_div.h?  divisor, data (checked by the translator) and every lane's mean inside its range;
  // either form returns the correctly rounded quotient, so which one runs never shows in the result
  __device__ __forceinline__ static bool all_mid(double mean, bool has) {
    const bool mine = !has || mean == 0 || mid_range(__builtin_fabs(mean));
    return __ballot(mine) == ~0ull;
  }

  // ---- log_post in the row layout: this lane's partial sum (the butterfly follows in amwg_kernel.h log_post)
  template <int U>
  __device__ __forceinline__ static double rows_eval(Cache &k, const StateView &S, const DataRef &d, const unsigned char *smem, int sub, int pitch, int wave) {
    load(k, smem, pitch, sub);
    const double start = M::template head<64>(S, d, smem, sub);
    update_inv(k, M::row_sd(S, d));
    const double sd = k.sd;
    const double mean = k.my_group >= 0 ? S(M::kRowBase + k.my_group) : 0.0;
    const bool fast = k.inv_fast && M::kRowDataMid && all_mid(mean, k.my_group >= 0);
    const double *tile = reinterpret_cast<const double *>(smem);
    const int rows = local_rows(M::kRowGroups);
    const int spitch = term_pitch(pitch);
    double *scratch = const_cast<double *>(tile) + (size_t)64 * pitch + 8 + (size_t)wave * rows * spitch;      // (+ 8 doubles: the 64 label bytes)
    const bool hitA = f64_bits(start) == f64_bits(k.a_start) && f64_bits(mean) == f64_bits(k.a_mean) && f64_bits(sd) == f64_bits(k.a_sd);
    const bool hitB = f64_bits(start) == f64_bits(k.b_start) && f64_bits(mean) == f64_bits(k.b_mean) && f64_bits(sd) == f64_bits(k.b_sd);
    const bool miss = !(hitA || hitB);
    const uint64_t missing = __ballot(miss);
    double T = hitA ? k.a_T : k.b_T;
    if (missing != 0ull) {
      double Tn;
      if (!fast || __popcll(missing) > rows) {
        Tn = rows_full<U>(k, fast, mean, tile + (size_t)sub * pitch, sub, start);
      } else {
        constexpr int n_full = M::kRowN >> 6, rem = M::kRowN & 63;
        // the terms of the stale lanes' observations, side by side (two stale lanes per trip, two rounds of 64 observations each in flight), left in
        // LDS; the stale lane then adds them up in order -- the same additions of the same values in the same order as its own pass
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
            const double e0a = k.c - div_by_invariant(t0a * t0a, k.den, k.y), e1a = k.c - div_by_invariant(t1a * t1a, k.den, k.y);
            const double e0b = k.c - div_by_invariant(t0b * t0b, k.den, k.y), e1b = k.c - div_by_invariant(t1b * t1b, k.den, k.y);
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
        if (miss) {
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

  // ---- sweep prefetch (amwg_kernel.h kSweep).  The stepper has drawn the proposals of a whole sweep over theta -- lane c holds the proposal of theta_c --
  // and asks for every lane's sum under the proposal of ITS component.  The translator has PROVED (kRowSweep) that a lane's head reads theta only as
  // theta[k] in a lane-split loop over all of theta (lane k: entry k) and that lane k < groups has label k: a lane's three numbers depend on ONE
  // entry of theta.  The proposals are therefore written into the chain's state all at once, the head and the mean are read off that state -- every
  // lane sees its own entry proposed, and whatever else was proposed does not reach it -- and the state is put back.
  template <int U>
  __device__ __forceinline__ static UserSweepRows prefetch_rows(Cache &k, const StateView &S, const ModelConsts &, const DataRef &d, const unsigned char *smem, int sub, double prop_own, int pitch) {
    UserSweepRows out{false, 0.0, 0.0, -1, false};
    if constexpr (M::kRowSweep) {
      load(k, smem, pitch, sub);
      update_inv(k, M::row_sd(S, d));
      const double sd = k.sd;
      const bool has = k.my_group >= 0;
      const double start_cur = M::template head<64>(S, d, smem, sub);
      const double mean_cur = has ? S(M::kRowBase + k.my_group) : 0.0;
      double *Sw = const_cast<double *>(S.base);
      double keep = 0.0;
      if (sub < M::kRowGroups) { keep = Sw[M::kRowBase + sub]; Sw[M::kRowBase + sub] = prop_own; }
      const double start_new = M::template head<64>(S, d, smem, sub);
      const double mean_new = has ? S(M::kRowBase + k.my_group) : 0.0;
      if (sub < M::kRowGroups) Sw[M::kRowBase + sub] = keep;
      if (!(k.inv_fast && M::kRowDataMid && all_mid(mean_cur, has) && all_mid(mean_new, has))) return out;      // (the IEEE-division pass: update by update)
      const double *row = reinterpret_cast<const double *>(smem) + (size_t)sub * pitch;
      bool curA = f64_bits(start_cur) == f64_bits(k.a_start) && f64_bits(mean_cur) == f64_bits(k.a_mean) && f64_bits(sd) == f64_bits(k.a_sd);
      const bool curB = f64_bits(start_cur) == f64_bits(k.b_start) && f64_bits(mean_cur) == f64_bits(k.b_mean) && f64_bits(sd) == f64_bits(k.b_sd);
      if (__ballot(!(curA || curB)) != 0ull) {      // the committed state's sum is not in the cache (the first sweep of a launch): formed now, by all lanes
        const double Tc = rows_full<U>(k, true, mean_cur, row, sub, start_cur);
        if (!(curA || curB)) { k.a_start = start_cur; k.a_mean = mean_cur; k.a_sd = sd; k.a_T = Tc; curA = true; }
      }
      out.T_cur = curA ? k.a_T : k.b_T;
      const double Tn = rows_full<U>(k, true, mean_new, row, sub, start_new);
      const bool intoB = curA;
      k.b_start = intoB ? start_new : k.b_start; k.b_mean = intoB ? mean_new : k.b_mean; k.b_sd = intoB ? sd : k.b_sd; k.b_T = intoB ? Tn : k.b_T;
      k.a_start = intoB ? k.a_start : start_new; k.a_mean = intoB ? k.a_mean : mean_new; k.a_sd = intoB ? k.a_sd : sd; k.a_T = intoB ? k.a_T : Tn;
      k.a_recent = intoB;      // (the committed one counts as recently used: a miss replaces the other)
      out.ok = true;
      out.T_new = Tn;
      out.comp = has ? k.my_group : (sub < M::kRowGroups ? sub : -1);
      out.new_in_b = intoB;
    }
    return out;
  }
  __device__ __forceinline__ static void sweep_done(Cache &k, const UserSweepRows &r, bool accepted_mine) { k.a_recent = r.new_in_b != accepted_mine; }

  // ---------------------------------------------------------------------------------------------------------------------------------
  // CERTIFIED DECISIONS in the row layout for a translated closure (round 6; the hand-written twin is HierNormalModel's, amwg_models.h; amwg_kernel.h
  // "certified decisions").  Models the translator marks kRowCert (the head only ever ADDS to the accumulator; the sweep is proved).
  // As a real number a lane's share of log_post is   head_l + n_l c - S2_l / den,   S2_l = sum over its row of (y_i - mean_l)^2 -- two operations per observation where
  // the term takes eight, and S2 depends on the lane's MEAN only: an update of a parameter outside the swept vector needs no pass.  The reference's expression
  // (reference_order below: the value these kernels decide against and leave behind) is ONE running sum: the head's K terms in the closure's order, then the n
  // observations.  Bounds, u = 2^-53: with H_l = the magnitudes of the head's terms dealt to lane l (M::head_pair), m_l = H_l + n_l |c| + Q_l and M = sum_l m_l,
  //     the running sum: K + n additions of partial sums below M, the terms' own roundings 5 u (n |c| + Q):      within (n + K + 8) u M of the real number
  //     the value here: the lane's head (its <= K terms in the lane's order: K u H_l), n_l c - Q_l as in the hand-written family ((n_l / 4 + 9) u m_l),
  //     the butterfly (6 u M):                                                                                    within (n_l / 4 + K + 15) u M
  // => a value: eps = u M (2 (n + K) + 64) 1.25,  a difference of two: u M (4 (n + K) + 128) 1.25 (M over max(m_l, m_l')).  value_bound / difference_bound have the
  // family's signature (M, d): the K-dependent factor is folded into the M that is handed to them -- M_eff = M (1 + 2 K / (2 n + 64)) gives exactly those numbers.
  // (the head's terms themselves are the same fp64 numbers in both orders: a term is the same operations on the same values whichever lane forms it)
  struct Approx { double value, eps; };
  struct ApproxLane { double value, mag; };
  __device__ __forceinline__ static double value_bound(double Mm, const DataRef &) { return Mm * (2.0 * (double)M::kRowN + 64.0) * 1.25 * 0x1p-53; }
  __device__ __forceinline__ static double difference_bound(double Mm, const DataRef &) { return Mm * (4.0 * (double)M::kRowN + 128.0) * 1.25 * 0x1p-53; }
  __device__ __forceinline__ static double head_factor(double K) { return 1.0 + 2.0 * K / (2.0 * (double)M::kRowN + 64.0); }
  // S2 of this lane's row for `mean`: four interleaved partial sums (the order is free: the value is used with its bound).
  // OUT OF LINE (round 6, last day): inlined at its three call sites -- each with its eight staged loads behind a scheduling fence -- the certified sweep kernel of a
  // translated closure needed ~780 vector registers and, in 512-thread workgroups (256 per lane), spilled 520 of them: 1 200 bytes of scratch per lane, ~250 scratch
  // accesses per step.  As a call it spills 40 (112 bytes); the call costs ~30 instructions beside the pass's ~320.  Translated cfg4: 1.87e9 -> 2.82e9 with the call,
  // 3.94e9 once the call's row argument became an LDS offset (below).
  // (the row travels as its LDS byte offset: across a call boundary a `const double *` is a generic pointer, and the pass read the tile with flat loads -- 9 000 cycles
  // per pass where ds_read takes 2 500)
  typedef __attribute__((address_space(3))) const double *LdsRow;
  __device__ __forceinline__ static uint32_t lds_offset(const double *p) { return (uint32_t)(uintptr_t)(LdsRow)p; }
  __device__ __attribute__((noinline)) static double rows_sq(uint32_t row_off, double mean, int sub) {
    const LdsRow row = (LdsRow)(uintptr_t)row_off;
    constexpr int n_full = M::kRowN >> 6, rem = M::kRowN & 63;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int r = 0;
    for (; r + 8 <= n_full; r += 8) {
      double x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = row[r + u];
      AMWG_STAGE_FENCE();
      { const double t = x[0] - mean; a0 = __builtin_fma(t, t, a0); } { const double t = x[1] - mean; a1 = __builtin_fma(t, t, a1); }
      { const double t = x[2] - mean; a2 = __builtin_fma(t, t, a2); } { const double t = x[3] - mean; a3 = __builtin_fma(t, t, a3); }
      { const double t = x[4] - mean; a0 = __builtin_fma(t, t, a0); } { const double t = x[5] - mean; a1 = __builtin_fma(t, t, a1); }
      { const double t = x[6] - mean; a2 = __builtin_fma(t, t, a2); } { const double t = x[7] - mean; a3 = __builtin_fma(t, t, a3); }
    }
    for (; r < n_full; ++r) { const double t = row[r] - mean; a0 = __builtin_fma(t, t, a0); }
    if (sub < rem) { const double t = row[n_full] - mean; a1 = __builtin_fma(t, t, a1); }
    return (a0 + a1) + (a2 + a3);
  }
  __device__ __forceinline__ static double lane_s2(Cache &k, const unsigned char *smem, double mean, int pitch, int sub) {      // (every lane takes part: a wavefront-uniform call)
    const bool stale = f64_bits(mean) != f64_bits(k.s2_mean);
    if (__ballot(stale) != 0ull) {
      const double v = rows_sq(lds_offset(reinterpret_cast<const double *>(smem) + (size_t)sub * pitch), mean, sub);
      if (stale) { k.s2 = v; k.s2_mean = mean; }
    }
    return k.s2;
  }
  __device__ __forceinline__ static ApproxLane approx_lane(const Cache &k, double start, double hmag, double s2, int sub) {
    const double n_l = (double)((M::kRowN >> 6) + (sub < (M::kRowN & 63) ? 1 : 0));
    const double q = s2 * k.y.hi, nc = n_l * k.c;
    return ApproxLane{(start + nc) - q, hmag + __builtin_fabs(nc) + q};
  }
  // THE REFERENCE'S ORDER: the closure's own evaluation with one lane per chain -- head<1> is the head's statements in sequence --, continued by ONE running sum over the
  // observations i = 64 r + lane whose terms the lanes compute side by side (the same operations on the same values as the one-lane kernel's loop).  Out of line, rare.
  __device__ inline __attribute__((noinline)) static double reference_order_sum(double acc, const double *row, int sub, double mean, double c, double den, double yh, double yl, bool fast) {
    constexpr int n_full = M::kRowN >> 6, rem = M::kRowN & 63;
    for (int r = 0; r <= n_full; ++r) {
      const int cnt = r < n_full ? 64 : rem;
      const double t = row[sub < cnt ? r : 0] - mean;
      const double tt = t * t;
      const double term = c - (fast ? div_by_invariant(tt, den, Reciprocal{yh, yl}) : tt / den);
      for (int l = 0; l < cnt; ++l) acc += lane_double(term, l);
    }
    return acc;
  }
  // (the head in sequence, out of line with the rest: its code -- the closure's loops over the parameter vectors, unrolled -- and its registers are not the hot path's)
  __device__ __attribute__((noinline)) static double head_in_sequence(const StateView S, const DataRef &d, const unsigned char *smem) { return M::template head<1>(S, d, smem, 0); }
  template <int G>
  __device__ __forceinline__ static double reference_order(Cache &k, const StateView &S, const ModelConsts &, const DataRef &d, const unsigned char *smem, int sub) {
    static_assert(G == 64, "the row layout: a chain on one wavefront");
    load(k, smem, d.pad, sub);
    update_inv(k, M::row_sd(S, d));
    const double head1 = head_in_sequence(S, d, smem);      // (every lane forms the same number)
    const double mean = k.my_group >= 0 ? S(M::kRowBase + k.my_group) : 0.0;
    const bool fast = k.inv_fast && M::kRowDataMid && all_mid(mean, k.my_group >= 0);
    return reference_order_sum(head1, reinterpret_cast<const double *>(smem) + (size_t)sub * d.pad, sub, mean, k.c, k.den, k.y.hi, k.y.lo, fast);
  }
  // log_post of the state as it stands, cheaply (the updates of parameters outside the swept vector, and the swept vector's when a sweep is walked update by update)
  template <int G, int BT>
  __device__ __forceinline__ static Approx log_post_approx(Cache &k, const StateView &S, const ModelConsts &, const DataRef &d, const unsigned char *smem, int sub) {
    if (d.pad <= 0) return Approx{0.0, __builtin_inf()};      // (wave-uniform: not the row layout -- the expression)
    load(k, smem, d.pad, sub);
    update_inv(k, M::row_sd(S, d));
    const HeadPair h = M::template head_pair<64>(S, d, smem, sub);
    const double mean = k.my_group >= 0 ? S(M::kRowBase + k.my_group) : 0.0;
    const double s2 = lane_s2(k, smem, mean, d.pad, sub);
    const ApproxLane a = approx_lane(k, h.value, h.mag, s2, sub);
    const double value = butterfly<1, 64>(a.value), Mv = butterfly<1, 64>(a.mag), K = butterfly<1, 64>(h.cnt);
    return Approx{value, value_bound(Mv * head_factor(K), d)};
  }
  // the sweep: every lane's value now and under its entry's proposal (lane c < groups holds the proposal of entry c).  kRowSweep (the translator's proof): a lane's
  // head and mean read the swept vector only as ITS entry, so all proposals are written into the state at once, read off it, and the state is put back.
  struct SweepApprox { bool ok; int comp; double cur, neu, mag, mean_new, s2_new; };
  __device__ __forceinline__ static SweepApprox sweep_approx(Cache &k, const StateView &S, const ModelConsts &, const DataRef &d, const unsigned char *smem, int sub, double prop_own) {
    SweepApprox out{false, -1, 0.0, 0.0, 0.0, 0.0, 0.0};
    if constexpr (M::kRowSweep) {
      if (d.pad <= 0) return out;      // (wave-uniform)
      load(k, smem, d.pad, sub);
      update_inv(k, M::row_sd(S, d));
      const bool has = k.my_group >= 0;
      const HeadPair h0 = M::template head_pair<64>(S, d, smem, sub);
      const double mean_cur = has ? S(M::kRowBase + k.my_group) : 0.0;
      double *Sw = const_cast<double *>(S.base);
      double keep = 0.0;
      if (sub < M::kRowGroups) { keep = Sw[M::kRowBase + sub]; Sw[M::kRowBase + sub] = prop_own; }
      const HeadPair h1 = M::template head_pair<64>(S, d, smem, sub);
      const double mean_new = has ? S(M::kRowBase + k.my_group) : 0.0;
      if (sub < M::kRowGroups) Sw[M::kRowBase + sub] = keep;
      const double s2_cur = lane_s2(k, smem, mean_cur, d.pad, sub);
      const double s2_new = rows_sq(lds_offset(reinterpret_cast<const double *>(smem) + (size_t)sub * d.pad), mean_new, sub);
      const ApproxLane c0 = approx_lane(k, h0.value, h0.mag, s2_cur, sub), c1 = approx_lane(k, h1.value, h1.mag, s2_new, sub);
      const double K = butterfly<1, 64>(h0.cnt > h1.cnt ? h0.cnt : h1.cnt);
      out.ok = true;
      out.comp = has ? k.my_group : (sub < M::kRowGroups ? sub : -1);
      out.cur = c0.value; out.neu = c1.value;
      out.mag = (c0.mag > c1.mag ? c0.mag : c1.mag) * head_factor(K);
      out.mean_new = mean_new; out.s2_new = s2_new;
    }
    return out;
  }
  // what the accepted entries leave behind in this lane: the S2 that goes with the new mean (the state itself is written by the stepper)
  __device__ __forceinline__ static void sweep_approx_commit(Cache &k, const SweepApprox &sa, uint64_t, bool mine, double, int, const DataRef &) {
    const bool grp = k.my_group >= 0 && mine;
    k.s2 = grp ? sa.s2_new : k.s2;
    k.s2_mean = grp ? sa.mean_new : k.s2_mean;
  }
};

}  // namespace amwg
#endif
