// amwg_kval.h -- exact fast-forward of a sequential fp64 sum whose terms take K distinct values (K <= 16): the generalisation of
// amwg_twoval.h that translated closures use for `for (i ...) lp += ld.pois(y[i], rate)` / `ld.binom(y[i], size, prob)` with a
// loop-invariant rate over small-integer data (distributions.js:240-248, 282-284) -- there the term is a function of y[i] alone, so
// the sum (...((acc + t_0) + t_1)...) + t_{N-1} has as many distinct addends as the data has distinct values.
//
// As in amwg_twoval.h: once acc is negative and all addends are, the magnitudes add; while |acc| stays inside one binade its ulp u is
// fixed and RN(|acc| + |c_k|) = |acc| + d_k u with d_k = |c_k| rounded to a multiple of u -- the same d_k for every addition of c_k in
// that binade, so the significand after m more observations is A + sum_k n_k(m) d_k in exact integer arithmetic, n_k = how often value k
// occurs among them (per-value prefix counts of the data, computed once on the host).  Bisection on m finds how far the sum can go
// before the significand would reach 2^53; the one addition that leaves the binade is a real fp64 add.  ~log2(N) binades instead of N
// additions, the same bits as the sequential loop.
//
// Ties: |c_k| exactly half-way between two multiples of u -- possible in ONE binade per addend, the one where u is twice the lowest set
// bit of c_k, i.e. within the first ~2^(t+2) terms for an addend with t trailing zero bits -- round by the parity of acc, which with more
// than two addends depends on the ORDER of the others, not on their counts (amwg_twoval.h has closed forms for two).  Such a binade is
// summed term by term; every other binade is fast-forwarded.  Exact for any data; only as fast as the data allows.
#pragma once
#include "amwg_twoval.h"

namespace amwg {

constexpr int kMaxKValues = 16;

// Data-only tables (translate.js kValuedTables; tests/host/kval_fuzz.cpp builds them the same way), W = n / 32 + 2 words per array:
//   tab: for value k = 0 .. K-1:  mask_k[W] (bit i & 31 of word i >> 5: observation i has value k), then pre_k[W] (occurrences among
//        the observations [0, 32 w));   idx: the value index of every observation, one byte each
struct KValData {
  const uint32_t *tab;
  const uint8_t *idx;
  int n;
};
AMWG_HD size_t k_valued_words(int n_obs) { return (size_t)n_obs / 32 + 2; }

template <int K>
AMWG_HD_SHARED double k_valued_sum(double acc, const double (&c)[K], const KValData &B) {
  static_assert(K >= 1 && K <= kMaxKValues, "k_valued_sum: 1 .. 16 distinct addends");
  const int N = B.n;
  const size_t W = k_valued_words(N);
  int i = 0;
  // the addend of observation i: selects over the K values (they live in registers; an indexed local array would live in scratch memory)
  auto addend = [&](int obs) -> double {
    const int k = (int)B.idx[obs];
    double v = c[0];
#pragma unroll
    for (int q = 1; q < K; ++q) v = k == q ? c[q] : v;
    return v;
  };
  auto step = [&](int obs) { acc = acc + addend(obs); };
  auto count_before = [&](int k, int m) -> uint32_t {
    const uint32_t *mask = B.tab + (size_t)(2 * k) * W, *pre = mask + W;
    const int w = m >> 5;
    return pre[w] + (uint32_t)__builtin_popcount(mask[w] & low_mask(m & 31));
  };
  const uint64_t kMant = 0x000fffffffffffffull, kHidden = 0x0010000000000000ull, kSat = ~0ull;
  int ek[K];
  uint64_t mk[K];
  int emax = 0;
  bool ok = true;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const uint64_t b = f64_bits(-c[k]);
    ek[k] = (int)(b >> 52);                         // sign bit clear iff the addend is negative
    mk[k] = (b & kMant) | kHidden;
    ok = ok && c[k] < 0 && ek[k] > 0 && ek[k] < 0x7ff;
    emax = ek[k] > emax ? ek[k] : emax;
  }
  // every addend negative, finite and normal; anything else (a rate at a bound: -inf terms, NaN, ...) is summed term by term
  if (!ok) {
    for (; i < N; ++i) step(i);
    return acc;
  }
  while (i < N) {
    const uint64_t ab = f64_bits(-acc);
    const int e = (int)(ab >> 52);                  // includes the sign bit of -acc: > 0x7ff when acc > 0
    if (!(e >= emax + 1 && e < 0x7ff)) { step(i); ++i; continue; }   // acc not yet negative / not yet 2x the largest addend / inf / NaN
    uint64_t A = (ab & kMant) | kHidden;            // |acc| = A * 2^(e - 1075)
    uint64_t d[K];
    bool tie = false, any = false;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int s = e - ek[k];                      // >= 1
      if (s >= 54) d[k] = 0;
      else {
        const uint64_t r = mk[k] & ((1ull << s) - 1ull), h = 1ull << (s - 1);
        tie = tie || r == h;
        d[k] = (mk[k] >> s) + (r > h ? 1u : 0u);
      }
      any = any || d[k] != 0;
    }
    if (tie) {      // a half-way addend in this binade: its rounding depends on the order of the others -- term by term until the binade is left
      do { step(i); ++i; } while (i < N && (int)(f64_bits(-acc) >> 52) == e);
      continue;
    }
    if (!any) break;                                // every addend is below half an ulp of acc: nothing changes any more
    const uint64_t limit = (1ull << 53) - A;        // the significand may grow by strictly less than this
    uint32_t base[K];
#pragma unroll
    for (int k = 0; k < K; ++k) base[k] = count_before(k, i);
    // n * d for n < 2^31 and d < 2^53, saturating (two 32 x 32 -> 64 products; anything >= 2^53 certainly reaches `limit` <= 2^52)
    auto scaled = [&](uint32_t n, uint64_t dk) -> uint64_t {
      const uint64_t hi = (uint64_t)n * (uint32_t)(dk >> 32);
      if (hi >> 21) return kSat;
      return (hi << 32) + (uint64_t)n * (uint32_t)dk;
    };
    auto growth = [&](int m) -> uint64_t {          // growth of the significand over the observations [i, m); kSat when it certainly reaches `limit`
      uint64_t T = 0;
      bool sat = false;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (d[k] == 0) continue;
        const uint64_t t = scaled(count_before(k, m) - base[k], d[k]);
        sat = sat || t >= limit;
        T += t < limit ? t : 0;                     // (each kept term < 2^52, K <= 16: no overflow)
      }
      return (sat || T >= limit) ? kSat : T;
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
    if (i < N) { step(i); ++i; }                    // the addition that leaves the binade: a real fp64 add
  }
  return acc;
}

}  // namespace amwg
