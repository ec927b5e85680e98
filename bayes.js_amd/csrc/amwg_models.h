// amwg_models.h -- the built-in log_post functors (the user's JS closure, mcmc.js:958-960,
// for the BASELINE.json model families).  Each model gives
//   prior(S, mc)            sequential sum of the prior terms, in the closure's order
//   begin(S, mc, d) -> Pass loop-invariant values of one pass over the data
//   term<FAST>(pass, i)     the i-th observation's log density
// and the generic log_post() in amwg_kernel.h adds them in the documented order.
// S(p) reads scalar component p of this chain's state.
#pragma once
#include "amwg_div.h"
#include "amwg_ld.h"
#include "amwg_types.h"

namespace amwg {

// ld.norm(v, 0|m, sd) with constant sd (a prior): c_sd - (v-m)^2 / (2*sd*sd)
__device__ __forceinline__ double norm_const_sd(double v, double m, double c_sd, double den) {
  const double t = v - m;
  return c_sd - (t * t) / den;
}

// ---------------------------------------------------------------------------------------------
// x_i ~ norm(mu, sigma); mu ~ norm(m0,s0); sigma ~ unif(a,b)              README.md:22-36
struct NormalModel {
  static constexpr bool kSplitPrior = false;
  static constexpr bool kUser = false, kHasFast = true, kOneLanePass = false;
  static constexpr int kDerived = 0;
  static constexpr int kMaxThreads = 1024;   // workgroup size cap (= __launch_bounds__: 128 VGPRs per lane)
  static constexpr int kUnroll = 8;   // independent terms in flight per lane (ILP across the division chains)
  struct Pass { double mu, c, den; Reciprocal y; bool fast; const double *x; };
  __host__ __device__ static size_t lds_bytes(int n_obs, int, int) { return (size_t)n_obs * 8; }
  __device__ static void stage(unsigned char *smem, const DataRef &d, int tid, int nt, int) {
    double *dst = reinterpret_cast<double *>(smem);
    for (int i = tid; i < d.n_obs; i += nt) dst[i] = d.x[i];
  }
  __device__ __forceinline__ static double prior(const StateView &S, const ModelConsts &mc, const DataRef &) {
    double lp = 0;
    lp += norm_const_sd(S(0), mc.m0, mc.c0, mc.den0);
    const double sigma = S(1);
    lp += (sigma < mc.ua || sigma > mc.ub) ? -kInf : mc.lunif;
    return lp;
  }
  __device__ __forceinline__ static Pass begin(const StateView &S, const ModelConsts &mc, const DataRef &,
                                               const unsigned char *smem) {
    Pass ps;
    ps.mu = S(0);
    const double sd = S(1);
    ps.c = norm_c(mc.neg_half_log_2pi, sd);
    ps.den = norm_den(sd);
    ps.y = make_reciprocal(ps.den);
    ps.fast = !mc.exact_division && mc.data_mid_range && mid_range(ps.den) &&
              (ps.mu == 0 || mid_range(__builtin_fabs(ps.mu)));
    ps.x = reinterpret_cast<const double *>(smem);
    return ps;
  }
  template <bool FAST>
  __device__ __forceinline__ static double term(const Pass &ps, int i) {
    const double t = ps.x[i] - ps.mu;
    const double tt = t * t;
    return ps.c - (FAST ? div_by_invariant(tt, ps.den, ps.y) : tt / ps.den);
  }
};

// ---------------------------------------------------------------------------------------------
// Exact fast-forward of a sequential sum whose terms take only two values.
//
// The beta-Bernoulli pass is  acc = (...((acc + t_0) + t_1)...) + t_{N-1}  with every t_i one of two NEGATIVE constants
// (log theta for x_i = 1, log(1-theta) for x_i = 0), each `+` rounding to nearest-even (mcmc.js log_post closure,
// distributions.js:228-230).  Once acc is negative the magnitudes add; while |acc| stays inside one binade
// [2^e, 2^(e+1)) its ulp u = 2^(e-52) is fixed, and  RN(|acc| + |c|) = |acc| + d_c * u  with  d_c = |c| rounded to a
// multiple of u -- the same d_c for every addition of c in that binade, unless |c| sits exactly half-way between two
// multiples (a tie, resolved by the parity of acc: then this binade is simply summed term by term).  So inside a binade
// the significand of acc after m more observations is  A + n0(m)*d0 + n1(m)*d1  in exact integer arithmetic, n1 = number
// of ones among them (prefix popcounts of the data, computed once on the host).  The code finds, by bisection on m, how
// far the sum can go before the significand would reach 2^53, jumps there, performs the ONE addition that leaves the
// binade with a real fp64 add (rounding on the coarser grid is the hardware's), and repeats: ~log2(N) binades instead of
// N additions, the same bits as the sequential loop (tests compare the two on the device, chain by chain).
struct BitData {
  const uint32_t *w;      // observation i = bit (i & 31) of w[i >> 5]
  const uint32_t *pre;    // pre[k] = number of ones among observations [0, 32k)
  // ties (see below): om1 / om0 mark the ones (zeros) whose immediately preceding run of zeros (ones) has odd length,
  // po1 / po0 are their prefix counts per word -- data-only, computed once on the host
  const uint32_t *om1, *po1, *om0, *po0;
  int n;
};
__device__ __forceinline__ uint32_t low_mask(int b) { return b ? (0xffffffffu >> (32 - b)) : 0u; }
__device__ __forceinline__ int ones_before(const BitData &B, int m) {
  const int k = m >> 5;
  return (int)(B.pre[k] + (uint32_t)__builtin_popcount(B.w[k] & low_mask(m & 31)));
}
__device__ __forceinline__ int odd_before(const uint32_t *om, const uint32_t *po, int m) {
  const int k = m >> 5;
  return (int)(po[k] + (uint32_t)__builtin_popcount(om[k] & low_mask(m & 31)));
}
// first observation >= i whose value is `sym` (N if none)
__device__ __forceinline__ int first_symbol(const BitData &B, int i, uint32_t sym) {
  const int nw = (B.n + 31) >> 5;
  int k = i >> 5;
  if (k >= nw) return B.n;
  uint32_t wt = (sym ? B.w[k] : ~B.w[k]) & ~low_mask(i & 31);
  while (wt == 0) { if (++k >= nw) return B.n; wt = sym ? B.w[k] : ~B.w[k]; }
  const int j = k * 32 + __builtin_ctz(wt);
  return j < B.n ? j : B.n;
}

// Ties.  |c| sits exactly half-way between two multiples of u in ONE binade per addend (where the bits of its
// significand below u are 100...0).  There RN(A + q + 1/2) goes to the even neighbour: up iff A + q is odd, and the
// result is even.  With A's parity p as the only state this is still closed form over a stretch of observations:
//   both addends tie       after the first addition A is even for good: increments q_c + (q_c & 1)
//   one ties (symbol t),   d_n even: parity only changes at t, so the first t rounds by (p + q_t) & 1, later ones by q_t & 1
//   the other (n) does not d_n odd:  every n flips the parity, every t resets it to even, so a later t rounds by the parity
//                                    of the run of n's right before it (data-only: the odd-run marks), the first one by
//                                    p plus the distance to it
__device__ inline double two_valued_sum(double acc, double l1, double l0, const BitData &B) {
  const int N = B.n;
  int i = 0;
  auto step = [&](int idx) { acc = acc + (((B.w[idx >> 5] >> (idx & 31)) & 1u) ? l1 : l0); };
  const uint64_t kMant = 0x000fffffffffffffull, kHidden = 0x0010000000000000ull;
  const uint64_t b1 = f64_bits(-l1), b0 = f64_bits(-l0);
  const int e1 = (int)(b1 >> 52), e0 = (int)(b0 >> 52);     // sign bit clear iff the addend is negative
  // both addends negative, finite and normal; anything else (theta at a bound, NaN, ...) is summed term by term
  if (!(l1 < 0 && l0 < 0 && e1 > 0 && e1 < 0x7ff && e0 > 0 && e0 < 0x7ff)) {
    for (; i < N; ++i) step(i);
    return acc;
  }
  const uint64_t m1 = (b1 & kMant) | kHidden, m0 = (b0 & kMant) | kHidden;
  const int emax = e1 > e0 ? e1 : e0;
  const uint64_t kSat = ~0ull;
  while (i < N) {
    const uint64_t ab = f64_bits(-acc);
    const int e = (int)(ab >> 52);                 // includes the sign bit of -acc: > 0x7ff when acc > 0
    if (!(e >= emax + 1 && e < 0x7ff)) { step(i); ++i; continue; }   // acc not yet negative / not yet 2x the larger addend / inf / NaN
    uint64_t A = (ab & kMant) | kHidden;           // |acc| = A * 2^(e - 1075), 2^52 <= A < 2^53
    // |c| = (q + r/u) u: d = q rounded by r against u/2; tie when r == u/2
    uint64_t q1, q0, d1, d0;
    bool tie1 = false, tie0 = false;
    {
      const int s = e - e1;                          // >= 1
      if (s >= 54) { q1 = 0; d1 = 0; } else { const uint64_t r = m1 & ((1ull << s) - 1ull), h = 1ull << (s - 1); q1 = m1 >> s; tie1 = r == h; d1 = q1 + (r > h ? 1u : 0u); }
    }
    {
      const int s = e - e0;
      if (s >= 54) { q0 = 0; d0 = 0; } else { const uint64_t r = m0 & ((1ull << s) - 1ull), h = 1ull << (s - 1); q0 = m0 >> s; tie0 = r == h; d0 = q0 + (r > h ? 1u : 0u); }
    }
    const uint64_t limit = (1ull << 53) - A;          // the significand may grow by strictly less than this
    const uint64_t p = A & 1ull;
    const int c1_i = ones_before(B, i);
    // T(m): growth of the significand over observations [i, m); kSat when it certainly reaches `limit`
    uint64_t mulA, mulB;           // per-observation weights of ones / zeros (without the tie corrections)
    int mode;                      // 0 plain, 1 both tie, 2 one tie + even d_n, 3 one tie + odd d_n
    uint32_t tsym = 0;             // the tying symbol in modes 2, 3
    uint64_t qt = 0;
    int jt = N;                    // mode 3: first tying symbol at or after i
    uint64_t up_first = 0;
    if (!tie1 && !tie0) { mode = 0; mulA = d1; mulB = d0; }
    else if (tie1 && tie0) { mode = 1; mulA = q1 + (q1 & 1ull); mulB = q0 + (q0 & 1ull); }
    else {
      tsym = tie1 ? 1u : 0u;
      qt = tie1 ? q1 : q0;
      const uint64_t dn = tie1 ? d0 : d1;
      mulA = tie1 ? q1 : d1;
      mulB = tie1 ? d0 : q0;
      if ((dn & 1ull) == 0) mode = 2;
      else {
        mode = 3;
        jt = first_symbol(B, i, tsym);
        up_first = (p + (uint64_t)((jt - i) & 1) + qt) & 1ull;
      }
    }
    if (mulA == 0 && mulB == 0 && mode == 0) break;   // the addends are below half an ulp of acc: nothing changes any more
    const uint64_t kA = mulA ? (limit - 1) / mulA : kSat, kB = mulB ? (limit - 1) / mulB : kSat;
    const uint32_t bit_i = (B.w[i >> 5] >> (i & 31)) & 1u;
    const uint32_t *om = tsym ? B.om1 : B.om0, *po = tsym ? B.po1 : B.po0;
    const int odd_j = (mode == 3 && jt < N) ? odd_before(om, po, jt + 1) : 0;
    const int cnt_j = (mode == 3 && jt < N) ? (tsym ? ones_before(B, jt + 1) : (jt + 1 - ones_before(B, jt + 1))) : 0;
    auto growth = [&](int m) -> uint64_t {
      const uint64_t n1 = (uint64_t)(ones_before(B, m) - c1_i), n0 = (uint64_t)(m - i) - n1;
      if (n1 > kA || n0 > kB) return kSat;
      uint64_t T = n1 * mulA + n0 * mulB;               // each product <= limit - 1 < 2^53
      if (mode == 1) {
        // the first addition rounds by the parity of A; every later one finds A even
        if (m > i) { const uint64_t qf = bit_i ? q1 : q0; T = T - (qf + (qf & 1ull)) + qf + ((p + qf) & 1ull); }
      } else if (mode == 2) {
        const uint64_t nt = tsym ? n1 : n0;
        if (nt) T += ((p + qt) & 1ull) + (nt - 1) * (qt & 1ull);
      } else if (mode == 3) {
        if (m > jt) {
          const uint64_t cnt = (uint64_t)((tsym ? ones_before(B, m) : (m - ones_before(B, m))) - cnt_j);      // tying symbols in (jt, m)
          const uint64_t odd = (uint64_t)(odd_before(om, po, m) - odd_j);                                      // ... after an odd run
          T += up_first + ((qt & 1ull) ? (cnt - odd) : odd);
        }
      }
      return T;
    };
    int lo = i, hi = N;
    uint64_t T_lo = 0, T = growth(N);
    if (T < limit) { lo = N; T_lo = T; }
    else {
      while (hi - lo > 1) {
        const int mid = lo + ((hi - lo) >> 1);
        T = growth(mid);
        if (T < limit) { lo = mid; T_lo = T; } else hi = mid;
      }
    }
    A += T_lo;
    acc = -bits_f64(((uint64_t)e << 52) | (A & kMant));
    i = lo;
    if (i < N) { step(i); ++i; }    // the addition that leaves the binade: a real fp64 add
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------
// x_i ~ bern(theta); theta ~ beta(a,b)                                    README.md:149-164
// ld.bern(x,p) = log(x*p + (1-x)*(1-p)) is exactly log(p) for x=1 and log(1-p) for x=0
// (1*p + 0*(1-p) = p + 0 = p), so the two logs are hoisted and selected per observation.
struct BetaBernModel {
  static constexpr bool kSplitPrior = false;
  static constexpr bool kUser = false, kHasFast = false, kOneLanePass = true;
  static constexpr int kDerived = 0;
  static constexpr int kMaxThreads = 1024;   // workgroup size cap (= __launch_bounds__: 128 VGPRs per lane)
  static constexpr int kUnroll = 8;
  struct Pass { double l1, l0; const uint8_t *x; const uint32_t *bits; bool has_invalid, fast_forward; BitData B; };
  // one lane per chain: the observations as bits plus their prefix popcounts in LDS (two_valued_sum above); the
  // term-by-term pass (exact_division = 1) reads the bits through the scalar cache instead
  __host__ __device__ static size_t words(int n_obs) { return (size_t)n_obs / 32 + 2; }
  static constexpr size_t kFfLdsLimit = 120 * 1024;   // beyond this the six bit arrays stay in HBM/L2 (read through L1)
  __host__ __device__ static size_t lds_bytes(int n_obs, int, int lanes) {
    if (lanes == 1) { const size_t b = 6 * words(n_obs) * 4; return b <= kFfLdsLimit ? ((b + 15) & ~(size_t)15) : 0; }
    return ((size_t)n_obs + 15) & ~(size_t)15;
  }
  __device__ static void stage(unsigned char *smem, const DataRef &d, int tid, int nt, int lanes) {
    if (lanes == 1) {   // d.arr[0]: the six arrays of BitData back to back (w, pre, om1, po1, om0, po0)
      if (lds_bytes(d.n_obs, 0, 1) == 0) return;
      uint32_t *dst = reinterpret_cast<uint32_t *>(smem);
      const uint32_t *src = static_cast<const uint32_t *>(d.arr[0]);
      for (int k = tid; k < (int)(6 * words(d.n_obs)); k += nt) dst[k] = src[k];
      return;
    }
    for (int i = tid; i < d.n_obs; i += nt) smem[i] = d.xb[i];
  }
  __device__ __forceinline__ static double prior(const StateView &S, const ModelConsts &mc, const DataRef &) {
    const double th = S(0);
    double lp = 0;
    if (th > 1 || th < 0) lp += -kInf;
    else if (mc.ba == 1 && mc.bb == 1) lp += 0.0;
    else lp += (mc.ba - 1) * log_v8(th) + (mc.bb - 1) * log_v8(1 - th) - mc.lbeta_ab;
    return lp;
  }
  __device__ __forceinline__ static Pass begin(const StateView &S, const ModelConsts &mc, const DataRef &d,
                                               const unsigned char *smem) {
    Pass ps;
    const double th = S(0);
    ps.l1 = log_v8(th);       // x = 1: log(1*th + 0*(1-th))
    ps.l0 = log_v8(1 - th);   // x = 0: log(0*th + 1*(1-th))
    ps.x = smem;
    ps.bits = d.xw;
    ps.has_invalid = mc.has_invalid != 0;
    ps.fast_forward = !mc.exact_division;
    {
      const size_t W = words(d.n_obs);
      const uint32_t *base = lds_bytes(d.n_obs, 0, 1) ? reinterpret_cast<const uint32_t *>(smem) : static_cast<const uint32_t *>(d.arr[0]);
      ps.B.w = base; ps.B.pre = base + W; ps.B.om1 = base + 2 * W; ps.B.po1 = base + 3 * W; ps.B.om0 = base + 4 * W; ps.B.po0 = base + 5 * W;
      ps.B.n = d.n_obs;
    }
    return ps;
  }
  template <bool FAST>
  __device__ __forceinline__ static double term(const Pass &ps, int i) { return ps.x[i] ? ps.l1 : ps.l0; }

  // Sequential sum for ONE lane per chain.  Every lane of the wave adds the same observation at
  // the same time, so the observation bit is wave-uniform: it is read through the scalar path
  // (32 observations per s_load'ed word) and selects, with a scalar branch, WHICH per-lane
  // register (log theta or log(1-theta)) the single v_add_f64 of that observation adds --
  // 1 vector instruction per observation instead of compare + 2 selects + add.  The adds are
  // inline asm so the compiler cannot turn the uniform branch back into per-lane selects.
  // Sequential sum for ONE lane per chain (the reference's order).  Every lane of the wave adds
  // the same observation at the same time, so the observation bit is wave-uniform: it travels
  // through the scalar unit and only decides WHICH per-lane register (log theta or
  // log(1-theta)) the observation's single v_add_f64 adds -- 1 vector instruction per
  // observation instead of compare + 2 selects + add.  On CDNA a SIMD issues at most one scalar
  // instruction per 4-cycle turn, the same cadence as one fp64 add, so a per-observation
  // test-and-branch (>= 3 scalar/branch issues) runs at 12 cycles per observation (measured).
  // Instead 8 observations are dispatched at once: the next data byte indexes a table of 256
  // straight-line blocks (8 adds + branch back, 68 bytes each) via s_setpc_b64 -- ~0.9 scalar
  // issues per observation, leaving the fp64 adds as the bound.  Hand-written because the
  // compiler would turn the uniform choice back into per-lane selects.
#define AMWG_A0 "v_add_f64 %[acc], %[acc], %[l0]\n"
#define AMWG_A1 "v_add_f64 %[acc], %[acc], %[l1]\n"
#define AMWG_N0 AMWG_A0 AMWG_A0 AMWG_A0 AMWG_A0
#define AMWG_N1 AMWG_A1 AMWG_A0 AMWG_A0 AMWG_A0
#define AMWG_N2 AMWG_A0 AMWG_A1 AMWG_A0 AMWG_A0
#define AMWG_N3 AMWG_A1 AMWG_A1 AMWG_A0 AMWG_A0
#define AMWG_N4 AMWG_A0 AMWG_A0 AMWG_A1 AMWG_A0
#define AMWG_N5 AMWG_A1 AMWG_A0 AMWG_A1 AMWG_A0
#define AMWG_N6 AMWG_A0 AMWG_A1 AMWG_A1 AMWG_A0
#define AMWG_N7 AMWG_A1 AMWG_A1 AMWG_A1 AMWG_A0
#define AMWG_N8 AMWG_A0 AMWG_A0 AMWG_A0 AMWG_A1
#define AMWG_N9 AMWG_A1 AMWG_A0 AMWG_A0 AMWG_A1
#define AMWG_N10 AMWG_A0 AMWG_A1 AMWG_A0 AMWG_A1
#define AMWG_N11 AMWG_A1 AMWG_A1 AMWG_A0 AMWG_A1
#define AMWG_N12 AMWG_A0 AMWG_A0 AMWG_A1 AMWG_A1
#define AMWG_N13 AMWG_A1 AMWG_A0 AMWG_A1 AMWG_A1
#define AMWG_N14 AMWG_A0 AMWG_A1 AMWG_A1 AMWG_A1
#define AMWG_N15 AMWG_A1 AMWG_A1 AMWG_A1 AMWG_A1
#define AMWG_B(HI, LO) AMWG_N##LO AMWG_N##HI "s_branch 25b\n"     /* byte HI*16+LO: low nibble = first 4 observations */
#define AMWG_ROW(HI)                                                                                   \
  AMWG_B(HI, 0) AMWG_B(HI, 1) AMWG_B(HI, 2) AMWG_B(HI, 3) AMWG_B(HI, 4) AMWG_B(HI, 5) AMWG_B(HI, 6) AMWG_B(HI, 7) \
  AMWG_B(HI, 8) AMWG_B(HI, 9) AMWG_B(HI, 10) AMWG_B(HI, 11) AMWG_B(HI, 12) AMWG_B(HI, 13) AMWG_B(HI, 14) AMWG_B(HI, 15)
#define AMWG_BERN_ADD_BIT0                                                                             \
  asm volatile("s_bitcmp1_b32 %3, 0\n\ts_cbranch_scc1 1f\n\tv_add_f64 %0, %0, %2\n\ts_branch 2f\n"        \
               "1:\n\tv_add_f64 %0, %0, %1\n2:"                                                        \
               : "+v"(acc) : "v"(l1), "v"(l0), "s"(w) : "scc")
  __device__ __forceinline__ static double pass_one_lane(const Pass &ps, int n_obs, double acc) {
    if (ps.fast_forward) return two_valued_sum(acc, ps.l1, ps.l0, ps.B);
    return pass_one_lane_sequential(ps, n_obs, acc);
  }
  __device__ __attribute__((noinline)) static double pass_one_lane_sequential(const Pass &ps, int n_obs, double acc) {
    const double l1 = ps.l1, l0 = ps.l0;
    // wave-uniform by construction; make that explicit so they live in SGPRs
    const uint64_t pbits = (uint64_t)reinterpret_cast<uintptr_t>(ps.bits);
    const uint32_t plo = __builtin_amdgcn_readfirstlane((uint32_t)pbits), phi = __builtin_amdgcn_readfirstlane((uint32_t)(pbits >> 32));
    const uint32_t *bits = reinterpret_cast<const uint32_t *>((uintptr_t)(((uint64_t)phi << 32) | plo));
    n_obs = __builtin_amdgcn_readfirstlane(n_obs);
    const int nw = n_obs >> 5;      // full 32-observation words go through the byte-dispatch loop
    if (nw > 0) {
      asm volatile(
          "s_mov_b64 s[40:41], %[ptr]\n"
          "s_mov_b32 s42, %[nw]\n"
          "s_getpc_b64 s[44:45]\n"
          "10:\n"
          "s_add_u32 s44, s44, 30f-10b\n"      // s[44:45] = address of the block table
          "s_addc_u32 s45, s45, 0\n"
          "20:\n"                              // ---- next word: s[46:47] = {word, sentinel 1}
          "s_load_dword s46, s[40:41], 0x0\n"
          "s_mov_b32 s47, 1\n"
          "s_add_u32 s40, s40, 4\n"
          "s_addc_u32 s41, s41, 0\n"
          "s_waitcnt lgkmcnt(0)\n"
          "22:\n"                              // ---- next byte of the word
          "s_and_b32 s48, s46, 0xff\n"
          "s_mul_i32 s48, s48, 68\n"           // 8 * 8-byte VOP3 adds + 4-byte s_branch
          "s_add_u32 s48, s44, s48\n"
          "s_addc_u32 s49, s45, 0\n"
          "s_setpc_b64 s[48:49]\n"
          "25:\n"                              // ---- blocks return here
          "s_lshr_b64 s[46:47], s[46:47], 8\n"
          "s_cmp_lg_u64 s[46:47], 1\n"         // only the sentinel left => word done
          "s_cbranch_scc1 22b\n"
          "s_sub_u32 s42, s42, 1\n"
          "s_cmp_lg_u32 s42, 0\n"
          "s_cbranch_scc1 20b\n"
          "s_branch 40f\n"
          "30:\n"
          AMWG_ROW(0) AMWG_ROW(1) AMWG_ROW(2) AMWG_ROW(3) AMWG_ROW(4) AMWG_ROW(5) AMWG_ROW(6) AMWG_ROW(7)
          AMWG_ROW(8) AMWG_ROW(9) AMWG_ROW(10) AMWG_ROW(11) AMWG_ROW(12) AMWG_ROW(13) AMWG_ROW(14) AMWG_ROW(15)
          "40:\n"
          : [acc] "+v"(acc)
          : [l1] "v"(l1), [l0] "v"(l0), [ptr] "s"(bits), [nw] "s"(nw)
          : "s40", "s41", "s42", "s44", "s45", "s46", "s47", "s48", "s49", "scc", "memory");
    }
    if (n_obs & 31) {               // ragged tail, one test-and-branch per observation
      uint32_t w = __builtin_amdgcn_readfirstlane(bits[nw]);
      for (int b = 0; b < (n_obs & 31); ++b) {
        AMWG_BERN_ADD_BIT0;
        w >>= 1;
      }
    }
    return acc;
  }
#undef AMWG_BERN_ADD_BIT0
#undef AMWG_ROW
#undef AMWG_B
};

// ---------------------------------------------------------------------------------------------
// y_i ~ norm(theta[g_i], sigma); theta_g ~ norm(mu,10); mu ~ norm(0,100); sigma ~ unif(0,100)
// components: theta[0..G-1], mu, sigma                                    SURVEY.md §8(d) cfg4
struct HierNormalModel {
  static constexpr bool kUser = false, kHasFast = true, kOneLanePass = false;
  static constexpr int kDerived = 0;
  static constexpr int kMaxThreads = 1024;   // workgroup size cap (= __launch_bounds__: 128 VGPRs per lane)
  static constexpr int kUnroll = 8;
  struct Pass { double c, den; Reciprocal y; bool fast; const double *x; const uint8_t *g; StateView S; };
  __host__ __device__ static size_t lds_bytes(int n_obs, int, int) { return (size_t)n_obs * 8 + (((size_t)n_obs + 15) & ~(size_t)15); }
  __device__ static void stage(unsigned char *smem, const DataRef &d, int tid, int nt, int) {
    double *dst = reinterpret_cast<double *>(smem);
    uint8_t *gd = smem + (size_t)d.n_obs * 8;
    for (int i = tid; i < d.n_obs; i += nt) { dst[i] = d.x[i]; gd[i] = d.xb[i]; }
  }
  // Lane order of the priors (the same a translated closure gets, translate.js): lane 0 adds the terms outside
  // loops -- mu, sigma -- and the `for k` loop over the group means is dealt to the lanes like the data loop.
  static constexpr bool kSplitPrior = true;
  __device__ __forceinline__ static double prior(const StateView &S, const ModelConsts &mc, const DataRef &d) {
    const double mu = S(d.G), sigma = S(d.G + 1);
    double lp = 0;
    lp += norm_const_sd(mu, mc.m0, mc.c0, mc.den0);
    lp += (sigma < mc.ua || sigma > mc.ub) ? -kInf : mc.lunif;
    return lp;
  }
  template <int G>
  __device__ __forceinline__ static double prior_split(const StateView &S, const ModelConsts &mc, const DataRef &d, int sub, double acc) {
    const double mu = S(d.G);
    for (int k = sub; k < d.G; k += G) acc += norm_const_sd(S(k), mu, mc.c1, mc.den1);
    return acc;
  }
  __device__ __forceinline__ static Pass begin(const StateView &S, const ModelConsts &mc, const DataRef &d,
                                               const unsigned char *smem) {
    Pass ps;
    const double sd = S(d.G + 1);
    ps.c = norm_c(mc.neg_half_log_2pi, sd);
    ps.den = norm_den(sd);
    ps.y = make_reciprocal(ps.den);
    bool ok = !mc.exact_division && mc.data_mid_range && mid_range(ps.den);
    for (int k = 0; k < d.G; ++k) { const double th = S(k); ok = ok && (th == 0 || mid_range(__builtin_fabs(th))); }
    ps.fast = ok;
    ps.x = reinterpret_cast<const double *>(smem);
    ps.g = smem + (size_t)d.n_obs * 8;
    ps.S = S;
    return ps;
  }
  template <bool FAST>
  __device__ __forceinline__ static double term(const Pass &ps, int i) {
    const double t = ps.x[i] - ps.S(ps.g[i]);
    const double tt = t * t;
    return ps.c - (FAST ? div_by_invariant(tt, ps.den, ps.y) : tt / ps.den);
  }
};

// ---------------------------------------------------------------------------------------------
// y_i ~ pois(exp(sum_k X[i][k]*beta[k] + [i >= cp]*beta[7])); beta_k ~ norm(0,10); cp ~ unif(0,N-1)
// components: beta[0..7], cp (int)                                        SURVEY.md §8(d) cfg5
// The design matrix (3.6 MB at N=5e4) is read straight from L2/MALL, stored column-major so the
// G lanes of a chain read consecutive observations of one column (coalesced); the per-observation
// exp+log dominate the arithmetic by two orders of magnitude.
struct PoisGlmModel {
  static constexpr bool kUser = false, kHasFast = false, kOneLanePass = false;
  static constexpr int kDerived = 0;
  static constexpr int kMaxThreads = 256;    // exp+log per observation want > 128 VGPRs; no LDS tile to share anyway
  static constexpr int kUnroll = 2;   // exp+log per term: more would spill
  struct Pass { double b[8]; double cp; const double *X, *y, *lfact; int N; };
  __host__ __device__ static size_t lds_bytes(int, int, int) { return 0; }
  __device__ static void stage(unsigned char *, const DataRef &, int, int, int) {}
  // closure order: `for k` over the 8 coefficients (dealt to the lanes), then the change point's prior (lane 0), then the data
  static constexpr bool kSplitPrior = true;
  __device__ __forceinline__ static double prior(const StateView &, const ModelConsts &, const DataRef &) { return 0.0; }
  template <int G>
  __device__ __forceinline__ static double prior_split(const StateView &S, const ModelConsts &mc, const DataRef &, int sub, double acc) {
    for (int k = sub; k < 8; k += G) acc += norm_const_sd(S(k), mc.m0, mc.c0, mc.den0);
    if (sub == 0) {
      const double cp = S(8);
      acc += (cp < 0 || cp > mc.cp_upper) ? -kInf : mc.lunif_cp;
    }
    return acc;
  }
  __device__ __forceinline__ static Pass begin(const StateView &S, const ModelConsts &, const DataRef &d,
                                               const unsigned char *) {
    Pass ps;
#pragma unroll
    for (int k = 0; k < 8; ++k) ps.b[k] = S(k);
    ps.cp = S(8);
    ps.X = d.x; ps.y = d.y; ps.lfact = d.lfact; ps.N = d.n_obs;
    return ps;
  }
  template <bool FAST>
  __device__ __forceinline__ static double term(const Pass &ps, int i) {
    double eta = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) eta += ps.X[(size_t)k * ps.N + i] * ps.b[k];
    if ((double)i >= ps.cp) eta += ps.b[7];
    const double lam = exp_v8(eta);
    return log_v8(lam) * ps.y[i] - lam - ps.lfact[i];
  }
};

}  // namespace amwg
