// amwg_twoval.h -- exact fast-forward of a sequential fp64 sum whose terms take only two values (used by the
// beta-Bernoulli functor of amwg_models.h and by translated closures whose data loop is `lp += ld.bern(x[i], p)`).
#pragma once
#include "amwg_math.h"
#include "amwg_types.h"

// large bodies: one copy per kernel, not one per call site (plain `inline`; amwg_math.h's AMWG_HD_OUTLINE is the noinline flavour)
#if defined(__HIPCC__)
#define AMWG_HD_SHARED __host__ __device__ inline
#else
#define AMWG_HD_SHARED inline
#endif

namespace amwg {

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
AMWG_HD uint32_t low_mask(int b) { return b ? (0xffffffffu >> (32 - b)) : 0u; }
AMWG_HD int ones_before(const BitData &B, int m) {
  const int k = m >> 5;
  return (int)(B.pre[k] + (uint32_t)__builtin_popcount(B.w[k] & low_mask(m & 31)));
}
AMWG_HD int odd_before(const uint32_t *om, const uint32_t *po, int m) {
  const int k = m >> 5;
  return (int)(po[k] + (uint32_t)__builtin_popcount(om[k] & low_mask(m & 31)));
}
// first observation >= i whose value is `sym` (N if none)
AMWG_HD int first_symbol(const BitData &B, int i, uint32_t sym) {
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
AMWG_HD_SHARED double two_valued_sum(double acc, double l1, double l0, const BitData &B) {
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
    // n * mul for n < 2^31 observations and mul < 2^53, saturating: (n * mul_hi) << 32 + n * mul_lo with two 32x32->64 products;
    // anything >= 2^53 certainly reaches `limit` (<= 2^52)
    const uint32_t aH = (uint32_t)(mulA >> 32), aL = (uint32_t)mulA, bH = (uint32_t)(mulB >> 32), bL = (uint32_t)mulB;
    auto scaled = [&](uint32_t n, uint32_t mh, uint32_t ml) -> uint64_t {
      const uint64_t hi = (uint64_t)n * mh;
      if (hi >> 21) return kSat;
      return (hi << 32) + (uint64_t)n * ml;       // < 2^53 + 2^63
    };
    const uint32_t bit_i = (B.w[i >> 5] >> (i & 31)) & 1u;
    const uint32_t *om = tsym ? B.om1 : B.om0, *po = tsym ? B.po1 : B.po0;
    const int odd_j = (mode == 3 && jt < N) ? odd_before(om, po, jt + 1) : 0;
    const int cnt_j = (mode == 3 && jt < N) ? (tsym ? ones_before(B, jt + 1) : (jt + 1 - ones_before(B, jt + 1))) : 0;
    auto growth = [&](int m) -> uint64_t {
      const uint32_t n1 = (uint32_t)(ones_before(B, m) - c1_i), n0 = (uint32_t)(m - i) - n1;
      const uint64_t t1 = scaled(n1, aH, aL), t0 = scaled(n0, bH, bL);
      if (t1 >= limit || t0 >= limit) return kSat;      // also catches the saturated products
      uint64_t T = t1 + t0;                              // < 2^54
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


// number of 32-bit words of each of the six tables for n observations
AMWG_HD size_t two_valued_words(int n_obs) { return (size_t)n_obs / 32 + 2; }

}  // namespace amwg
