// amwg_ptail.h -- CERTIFIED POISSON TAIL of a translated closure (round 6; bayes.js_amd/translate.js poisTailPlan; amwg_kernel.h "certified decisions").
//
// A closure whose last statement is
//     for (i = 0; i < N; i++) { <statements forming eta from the state and row i of the data>;  lp += ld.pois(y[i], Math.exp(eta)); }
// -- a log-link count regression, whatever the linear predictor looks like (amwg_models.h PoisGlmModel is the hand-written instance).  The reference's term is
// log(lambda) y - lambda - lfactorial(y), lambda = exp(eta) (distributions.js:282-284): it takes the LOGARITHM of the exponential it has just formed, ~70 of the
// term's operations.  As real numbers   log_post = head + sum eta_i y_i - sum e^eta_i - sum lfactorial(y_i),   and that is what the pass below forms:
//   * eta_i by the closure's OWN statements (M::ptail_eta: the same operations in the same order as the expression's pass -- bit for bit the reference's eta_i);
//   * e^eta by exp_bounded (amwg_math.h: 17 operations, relative error < 2^-46), two running sums per chain; the third sum is a constant of the data;
//   * and it is the WAVEFRONT's pass (16 lanes per chain, four chains to a wavefront): the 64 lanes share out the OBSERVATIONS whichever chain they belong to, a row
//     -- once in registers -- is evaluated for all four chains, whose parameters are read from their LDS state at wave-uniform addresses and kept in SCALAR
//     registers (ScalarState: the translator has proved that every state index in the statements is the same for all observations).  A quarter of the memory
//     traffic of the expression's pass, which re-reads every row for every chain.
// The stepper gets the value with a bound eps on its distance from the expression evaluated in the REFERENCE's order (pois_tail_reference below: what this kernel
// evaluates when a uniform falls inside the bound, and what a launch leaves behind): u = 2^-53, H = max_i |eta_i| (taken over the rows as they pass: the etas are the
// reference's own), Y = sum y_i, F = sum lfactorial(y_i) (both formed by the translator, compensated), L = sum e^eta_i, Hm / Hc = the magnitudes / the number of the
// head's additions (M::ptail_head: HeadPair), W = Hm + (1 + H) Y + L + F >= every partial sum of magnitudes on either side:
//   the terms: log_v8(exp_v8(eta)) against eta: 2 u (1 + H) 1.01 per unit of y;  exp_v8 against e^eta: 2 u L;  exp_bounded: 2^-46 L = 128 u L;  the term's three
//   roundings: 4 u (H Y + L + F) -- together < 136 u W;  the reference's ONE running sum over the head's Hc terms and the n observations': (Hc + n) u W;  this pass:
//   the head in the lanes' order Hc u Hm, per-lane sums of n / 64 + 1 fused steps and six butterfly additions on two sums (n / 64 + 7) u (H Y + L), their difference
//   and the closing (P + tot) - F: 3 u W;  F itself: 2 u F.   In all  < u W (n + n / 64 + 2 Hc + 150);  the bound handed on is
//       eps = u W (n + n / 32 + 2 Hc + 23 H + 200) 1.25
//   (the hand-written family's formula is the same count with its nine prior terms; its 23 H term covers a DIFFERENT eta on the two sides: here it is slack.)
// LINEAR PREDICTORS (M::kTailLinear: eta is a sum of (row entry) x (state entry) products, state entries and literals -- the translator's proof on the loop's
// statements): the pass forms eta by fused steps, K + 1 roundings where the closure's statements have 2 K + 1, and H = sum |state entry| x max_i |row entry| (column
// maxima from the translator) bounds every summand's magnitude, hence |eta| and each rounding of either eta: |eta_fused - eta_reference| <= R u H 1.05, R the
// roundings of both; that distance enters sum eta y and sum e^eta as R u H (Y + L) 1.05 -- the coefficient of H in eps is then 1.1 R + 1 (the family: 13 + 7 -> 23).
// 36 operations per observation and chain against the generic 44, and no running maximum.
// H > 690 (exp and log leave their ordinary range), any non-finite value: eps is not finite and the stepper evaluates the expression.  A negative count makes the
// reference's term -inf: the translator does not emit this plan for such data.
// Checked like the other bounds: tools/bound_audit.py case user_pois_glm_closure (libamwg_audit.so evaluates the expression beside every certified value).
#pragma once
#include "amwg_user.h"      // (which includes this file at its end: TailApprox, ld_pois_pre_exp)
#if defined(__HIPCC__) || defined(__HIPCC_RTC__)      // (device code throughout: the host build of a generated model -- tests/host -- sees nothing of it)
#include "amwg_kernel.h"    // butterfly
#include "amwg_math.h"
#include "amwg_rows.h"      // HeadPair
#include "amwg_types.h"

namespace amwg {

// a chain's state in registers: read once per pass at wave-uniform LDS addresses, every value moved to a scalar register pair (M::kTailUniformState: the loop's
// statements index it by constants only, once their inner loops are unrolled)
template <int P>
struct ScalarState {
  double v[P];
  __device__ __forceinline__ double operator()(int p) const { return v[p]; }
};
__device__ __forceinline__ double wave_uniform(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint64_t v = f64_bits(x);
  return bits_f64(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
#else
  return x;
#endif
}

template <class M, int G, int BT>
__device__ __forceinline__ TailApprox pois_tail_approx(const StateView &S, const DataRef &d, const unsigned char *smem, int sub) {
  static_assert(G == 16, "the certified Poisson tail runs four chains to a wavefront");
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int CW = 64 / G;
  const int lane = (int)(threadIdx.x & 63u);
  const HeadPair h = M::template ptail_head<G>(S, d, smem, sub);
  const double P = butterfly<1, G>(h.value), Hm = butterfly<1, G>(h.mag), Hc = butterfly<1, G>(h.cnt);
  // the four chains' states: LDS addresses, wave-uniform (the first lane of each chain's)
  typedef __attribute__((address_space(3))) const double *lds_f64;
  const uint32_t mine_off = (uint32_t)(uintptr_t)(lds_f64)S.base;
  const double *base[CW];
#pragma unroll
  for (int c = 0; c < CW; ++c) base[c] = (const double *)(lds_f64)(uintptr_t)(uint32_t)__builtin_amdgcn_readlane((int)mine_off, c * G);
  const ExpTaylorRegs E = exp_taylor_regs();
  double s1[CW], ls[CW], hm = 0.0;
#pragma unroll
  for (int c = 0; c < CW; ++c) { s1[c] = 0.0; ls[c] = 0.0; }
  constexpr int n = M::kTailN;
  // a LINEAR predictor (M::kTailLinear; needs the row cache): eta by fused steps -- one rounding per product where the closure's statements have two --, and
  // H = M::ptail_hlin(S) = sum |state entry| x column maximum >= sum of the magnitudes of eta's summands, once per pass, instead of max |eta_i| over the rows
  constexpr bool kLinear = M::kTailLinear && M::kTailRows;
  auto pass = [&](const auto &Sc) {
    auto consume = [&](const double (&eta)[CW], double y) {
      double lam[CW];
      if constexpr (!kLinear) {
#pragma unroll
        for (int c = 0; c < CW; ++c) hm = __builtin_fmax(hm, __builtin_fabs(eta[c]));
      }
#pragma unroll
      for (int c = 0; c < CW; ++c) lam[c] = exp_bounded(eta[c], E);
#pragma unroll
      for (int c = 0; c < CW; ++c) { s1[c] = __builtin_fma(eta[c], y, s1[c]); ls[c] += lam[c]; }
    };
    if constexpr (M::kTailRows) {
      // a row is loaded a round AHEAD of its use (M::ptail_load: the translator has proved that the statements read nothing of the data but the observation's own
      // row): the loads of round k + 1 are in flight while round k's four linear predictors and exponentials (~175 instructions) run.  Two row buffers, alternating.
      // (measured on cfg5's closure, 8 192 chains: 1.397e7 updates/s against 1.366e7 for the plain loop below; forming the next round's linear predictors ahead
      // instead -- no row buffers -- was slower than either, 1.32e7: the compiler waits for the loads right behind their issue)
      constexpr int n_full = n / 64, rem = n % 64, pairs = n_full > 0 ? (n_full - 1) / 2 : 0, left = n_full - 2 * pairs;      // left: 0 (no full round), 1 or 2
      auto compute = [&](const typename M::TailRow &R, int i) {
        double eta[CW];
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          if constexpr (kLinear) eta[c] = M::ptail_eta_fused(Sc[c], R, i);
          else eta[c] = M::ptail_eta_row(Sc[c], R, i);
        }
        consume(eta, M::ptail_y_row(R));
      };
      typename M::TailRow A, B;
      if constexpr (n_full > 0) {
        M::ptail_load(d, smem, lane, A);
        int i = lane;
        for (int k = 0; k < pairs; ++k, i += 128) {
          M::ptail_load(d, smem, i + 64, B);
          AMWG_STAGE_FENCE();
          compute(A, i);
          AMWG_STAGE_FENCE();
          M::ptail_load(d, smem, i + 128, A);
          AMWG_STAGE_FENCE();
          compute(B, i + 64);
          AMWG_STAGE_FENCE();
        }
        if constexpr (left == 2) {
          M::ptail_load(d, smem, i + 64, B);
          AMWG_STAGE_FENCE();
          compute(A, i);
          AMWG_STAGE_FENCE();
          compute(B, i + 64);
        } else {
          compute(A, i);
        }
      }
      if constexpr (rem > 0) {
        if (lane < rem) {
          M::ptail_load(d, smem, n_full * 64 + lane, A);
          compute(A, n_full * 64 + lane);
        }
      }
    } else {      // (some read of the data is not of the observation's own row: the plain loop -- every row is waited for where its first product needs it)
#pragma unroll 2
      for (int i = lane; i < n; i += 64) {
        const double y = M::ptail_y(d, smem, i);
        double eta[CW];
#pragma unroll
        for (int c = 0; c < CW; ++c) eta[c] = M::ptail_eta(Sc[c], d, smem, i);
        consume(eta, y);
      }
    }
  };
  if constexpr (M::kTailUniformState) {
    ScalarState<M::kStateN> Sc[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) {
#pragma unroll
      for (int p = 0; p < M::kStateN; ++p) Sc[c].v[p] = wave_uniform(base[c][p]);
    }
    pass(Sc);
  } else {
    StateView Sc[CW];      // (per-lane LDS reads: the loop gathers from the state by the data)
#pragma unroll
    for (int c = 0; c < CW; ++c) Sc[c].base = base[c];
    pass(Sc);
  }
  // every chain's totals over the wavefront; a lane keeps its own chain's.  H: the largest |eta| any of the four chains met (fmax skips a NaN: the sums carry it)
  double cH = 23.0;      // (slack for the closure's own eta on both sides; the hand-written family's coefficient)
  if constexpr (kLinear) {
    hm = M::ptail_hlin(S);      // this lane's own chain
    cH = 1.1 * (double)M::kTailLinearRoundings + 1.0;      // eta's roundings on the two sides, each below u H: |eta_fused - eta_reference| <= kTailLinearRoundings u H 1.05
  } else {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) hm = __builtin_fmax(hm, __shfl_xor(hm, o));
  }
  const int mine = lane / G;
  double tot = 0.0, L = 0.0;
#pragma unroll
  for (int c = 0; c < CW; ++c) {
    const double t = butterfly<1, 64>(s1[c] - ls[c]), l = butterfly<1, 64>(ls[c]);
    tot = mine == c ? t : tot;
    L = mine == c ? l : L;
  }
  const double Y = M::ptail_sum_y(), F = M::ptail_sum_lf(), H = hm;
  const double W = Hm + (1.0 + H) * Y + L + __builtin_fabs(F);
  const double eps = (H <= 690.0) ? W * ((double)n + (double)(n / 32) + 2.0 * Hc + cH * H + 200.0) * 1.25 * 0x1p-53 : __builtin_inf();
  return TailApprox{(P + tot) - F, eps};
#else
  (void)S; (void)d; (void)smem; (void)sub;
  return TailApprox{0.0, __builtin_inf()};
#endif
}

// THE REFERENCE'S ORDER at G lanes per chain: the head as the closure states it (one lane's walk: M::ptail_head_sequence), then ONE running sum over the observations'
// terms -- the chain's lanes form the terms of a round of G observations side by side (the expression's own operations: ld_pois_pre_exp of the closure's eta) and
// the sum takes them in the order i = G k + lane, a broadcast per term.  Slow (~1 ms for 5e4 observations), and run for ~1e-6 of the updates.  Under the chain's own
// execution mask: the lanes it reads are its own.
template <class M, int G>
__device__ inline __attribute__((noinline)) double pois_tail_reference(const double *state, const DataRef *dp, const unsigned char *smem, int sub) {
  double acc = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
  const StateView S{state};
  const DataRef &d = *dp;
  acc = M::ptail_head_sequence(S, d, smem);
  constexpr int n = M::kTailN;
  const int base = (int)(threadIdx.x & 63u) & ~(G - 1);
  for (int k0 = 0; k0 < n; k0 += G) {
    const int cnt = n - k0 < G ? n - k0 : G, i = sub < cnt ? k0 + sub : k0;
    const double term = ld_pois_pre_exp(M::ptail_y(d, smem, i), M::ptail_eta(S, d, smem, i), M::ptail_lf(d, smem, i));
    for (int l = 0; l < cnt; ++l) acc += __shfl(term, base + l, 64);
  }
#else
  (void)state; (void)dp; (void)smem; (void)sub;
#endif
  return acc;
}

}  // namespace amwg
#endif
