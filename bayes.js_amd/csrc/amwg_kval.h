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
// Ties: |c_k| exactly half-way between two multiples of u -- in the one binade per addend where u is twice the lowest set bit of c_k; an
// addend with s significand bits below u ties with probability 2^-s, so the binades right above the addends' own meet one often and a high
// binade now and then -- round by the parity of acc, which with more than two addends depends on the ORDER of the others, not on their
// counts (amwg_twoval.h has closed forms for two).  A binade with one tying addend is walked 32 observations at a time with bit tricks on
// the occurrence masks (below); one with several, term by term; every other binade is fast-forwarded.  Exact for any data.
#pragma once
#include "amwg_twoval.h"

namespace amwg {

constexpr int kMaxKValues = 16;

// Data-only tables (translate.js kValuedTables; tests/host/kval_fuzz.cpp builds them the same way), W = n / 32 + 2 blocks of 32 observations:
//   tab: for block w:  pre[K] (occurrences of every value among the observations [0, 32 w)), then mask[K] (bit j of mask[k]: observation
//        32 w + j has value k) -- the 2 K words of a block side by side: one evaluation of the bisection reads ONE block (a cache line or two; with
//        one array per value it read 2 K lines, and the loop was memory bound);   idx: the value index of every observation, one byte each
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
  (void)W;
  auto count_before = [&](int k, int m) -> uint32_t {
    const uint32_t *blk = B.tab + (size_t)(m >> 5) * (2 * K);
    return blk[k] + (uint32_t)__builtin_popcount(blk[K + k] & low_mask(m & 31));
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
  // Shape of the loop (it matters on the device, where the 64 lanes of a wavefront are 64 different chains): every trip of the outer loop
  // first lets each lane add term by term for as long as IT has to -- cheap trips, lanes wait for the slowest -- and then makes ONE
  // bisection for all lanes together.  (Written as one loop with the term-by-term cases as `continue`, the lanes drifted apart and every
  // trip paid a full bisection because SOME lane needed one: ten times the loads and instructions of a single chain.)
  while (i < N) {
    uint64_t A = 0, d[K];
    int e = 0;
    bool ready = false, any = false;
    while (i < N) {
      const uint64_t ab = f64_bits(-acc);
      e = (int)(ab >> 52);                          // includes the sign bit of -acc: > 0x7ff when acc > 0
      if (!(e >= emax + 1 && e < 0x7ff)) { step(i); ++i; continue; }   // acc not yet negative / not yet 2x the largest addend / inf / NaN
      A = (ab & kMant) | kHidden;                   // |acc| = A * 2^(e - 1075)
      int n_tie = 0, t_tie = 0;
      any = false;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int s = e - ek[k];                    // >= 1
        if (s >= 54) d[k] = 0;
        else {
          const uint64_t r = mk[k] & ((1ull << s) - 1ull), h = 1ull << (s - 1);
          if (r == h) { ++n_tie; t_tie = k; }
          d[k] = (mk[k] >> s) + (r > h ? 1u : 0u);  // (a tying addend: d = its floor q)
        }
        any = any || d[k] != 0;
      }
      if (n_tie != 0) {
        // A half-way addend in this binade: RN(A + q + 1/2) goes to the even neighbour -- up iff A + q is odd, and the result is even -- so its
        // rounding depends on the ORDER of the other addends (each flips the parity of A iff its d is odd), not on their counts.  With ONE
        // tying addend t the binade is still walked 32 observations at a time on the occurrence masks: with Z = the observations whose
        // addend has an odd d and X = the exclusive prefix XOR of Z, the parity in front of an occurrence of t is X there XOR X at the
        // previous occurrence (the parity is 0 right after one), or the incoming parity XOR X before the first; a fill-forward of X over
        // the occurrences of t gives all 32 at once.  (A lane of a wavefront that met such a binade term by term held the other 63 up for
        // thousands of additions: the longest of 64 such runs, not the average, was what an evaluation cost.)  Two or more tying addends,
        // and the block in which the binade is left: term by term.
        if (n_tie == 1) {
          const int t = t_tie;
          bool moved = false;
          while (i < N) {
            const int w = i >> 5;
            const uint32_t *blk = B.tab + (size_t)w * (2 * K);
            const uint32_t keep = ~low_mask(i & 31);
            uint32_t Tm = 0, Z = 0;
            uint64_t add = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
              const uint32_t m = blk[K + k] & keep;
              add += (uint64_t)__builtin_popcount(m) * d[k];
              if (k == t) Tm = m; else Z |= (d[k] & 1ull) ? m : 0u;
            }
            uint32_t x = Z;                          // inclusive prefix XOR towards later observations
            x ^= x << 1; x ^= x << 2; x ^= x << 4; x ^= x << 8; x ^= x << 16;
            const uint32_t X = x << 1;               // exclusive: bit j = parity of the odd addends among the block's observations before j
            uint32_t v = X & Tm, m = Tm;             // fill-forward: v = X at the last occurrence of t at or before each position (where m says there is one)
            v |= (v << 1) & ~m; m |= m << 1;
            v |= (v << 2) & ~m; m |= m << 2;
            v |= (v << 4) & ~m; m |= m << 4;
            v |= (v << 8) & ~m; m |= m << 8;
            v |= (v << 16) & ~m; m |= m << 16;
            const uint32_t prev_val = v << 1, prev_cov = m << 1;      // ... strictly before each position
            const uint32_t QB = (d[t] & 1ull) ? ~0u : 0u, PIN = (A & 1ull) ? ~0u : 0u;
            const uint32_t up = Tm & ((prev_cov & (X ^ prev_val ^ QB)) | (~prev_cov & (X ^ PIN ^ QB)));
            const uint64_t A_new = A + add + (uint64_t)__builtin_popcount(up);
            if (A_new >= (1ull << 53)) break;        // the binade is left inside this block
            A = A_new;
            i = (w + 1) * 32 < N ? (w + 1) * 32 : N;
            moved = true;
          }
          if (moved) acc = -bits_f64(((uint64_t)e << 52) | (A & kMant));
        }
        while (i < N && (int)(f64_bits(-acc) >> 52) == e) { step(i); ++i; }
        continue;
      }
      ready = true;
      break;
    }
    if (!ready || !any) break;                      // the data is used up | every addend is below half an ulp of acc: nothing changes any more
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
      uint32_t cnt[K];
#pragma unroll
      for (int k = 0; k < K; ++k) cnt[k] = count_before(k, m);      // (the 2 K words of block m >> 5: requested together, no branch in between)
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const uint64_t t = scaled(cnt[k] - base[k], d[k]);
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
