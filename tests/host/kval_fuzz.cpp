// kval_fuzz.cpp -- TEST INFRASTRUCTURE: the exact fast-forward of K-valued sequential sums (csrc/amwg_kval.h, the same source the
// kernels compile) against the plain fp64 loop, on the host, over random and adversarial cases (addends with trailing zero bits tie
// in reachable binades; powers of two; -inf / NaN / positive addends; sums that start positive).
//   g++ -std=c++17 -O2 -ffp-contract=off -I bayes.js_amd/csrc tests/host/kval_fuzz.cpp -o kval_fuzz && ./kval_fuzz [cases]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "amwg_kval.h"

using namespace amwg;

struct Tables { std::vector<uint32_t> tab; std::vector<uint8_t> idx; };
static Tables tables(const std::vector<uint8_t> &v, int K) {      // same recipe as translate.js kValuedTables
  const int N = (int)v.size();
  const size_t W = k_valued_words(N);
  Tables t;
  t.tab.assign((size_t)2 * K * W, 0u);
  t.idx = v;
  for (int i = 0; i < N; ++i) t.tab[((size_t)i >> 5) * (2 * K) + K + v[i]] |= 1u << (i & 31);
  for (size_t w = 1; w < W; ++w)
    for (int k = 0; k < K; ++k) t.tab[w * (2 * K) + k] = t.tab[(w - 1) * (2 * K) + k] + (uint32_t)__builtin_popcount(t.tab[(w - 1) * (2 * K) + K + k]);
  return t;
}

template <int K>
static long run(long cases, uint64_t seed, long *forced) {
  std::mt19937_64 rng(seed);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  long bad = 0;
  for (long c = 0; c < cases; ++c) {
    const int N = (c % 7 == 0) ? (int)(rng() % 30000) : (int)(rng() % 900);
    std::vector<double> w(K);
    double tot = 0;
    for (auto &x : w) { x = (c % 4 == 0) ? U(rng) * U(rng) * U(rng) : U(rng); tot += x; }
    std::vector<uint8_t> v((size_t)N);
    for (auto &x : v) { double u = U(rng) * tot; int k = 0; while (k < K - 1 && u > w[k]) { u -= w[k]; ++k; } x = (uint8_t)k; }
    const Tables t = tables(v, K);
    const KValData B{t.tab.data(), t.idx.data(), N};
    double cs[K];
    auto trail = [&](double x, int z, bool setbit) { uint64_t u; memcpy(&u, &x, 8); u = (u >> z) << z; if (setbit) u |= 1ull << z; memcpy(&x, &u, 8); return x; };
    for (int k = 0; k < K; ++k) {
      cs[k] = -std::exp(U(rng) * 15 - 12);
      if (c % 3 != 0) { cs[k] = trail(cs[k], (int)(rng() % 40), rng() & 1); }
    }
    if (c % 3 != 0) ++*forced;
    if (c % 97 == 0) cs[0] = -1.0;
    if (c % 89 == 0) cs[K - 1] = -0.5;
    if (c % 1013 == 0) cs[0] = -INFINITY;
    if (c % 1019 == 0) cs[K - 1] = NAN;
    if (c % 1021 == 0) cs[0] = 0.25;
    if (c % 53 == 0) for (int k = 1; k < K; ++k) cs[k] = cs[0] * (double)(k + 1);      // (a log-linear family: exact multiples tie together)
    double acc0 = (c % 2) ? -std::exp(U(rng) * 30 - 5) : (U(rng) - 0.5) * 6;
    if (c % 211 == 0) acc0 = 0.0;
    const double got = k_valued_sum<K>(acc0, cs, B);
    double want = acc0;
    for (int i = 0; i < N; ++i) want = want + cs[v[i]];
    if (memcmp(&got, &want, 8) != 0 && !(got != got && want != want)) {
      if (bad < 5) printf("MISMATCH K=%d case %ld N=%d acc0=%a got=%a want=%a\n", K, c, N, acc0, got, want);
      ++bad;
    }
  }
  return bad;
}

int main(int argc, char **argv) {
  const long cases = argc > 1 ? atol(argv[1]) : 6000;
  long forced = 0, bad = 0;
  bad += run<1>(cases / 4, 1, &forced);
  bad += run<2>(cases, 2, &forced);
  bad += run<3>(cases, 3, &forced);
  bad += run<5>(cases, 5, &forced);
  bad += run<8>(cases, 8, &forced);
  bad += run<16>(cases / 2, 16, &forced);
  printf("cases_per_K=%ld forced_trailing_zero_addends=%ld mismatches=%ld\n", cases, forced, bad);
  return bad ? 1 : 0;
}
