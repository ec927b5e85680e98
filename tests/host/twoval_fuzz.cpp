// twoval_fuzz.cpp -- TEST INFRASTRUCTURE: the exact fast-forward of two-valued sequential sums (csrc/amwg_twoval.h, the same
// source the kernels compile) against the plain fp64 loop, on the host, over random and adversarial cases.
//   g++ -std=c++17 -O2 -ffp-contract=off -I bayes.js_amd/csrc tests/host/twoval_fuzz.cpp -o twoval_fuzz && ./twoval_fuzz [cases]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "amwg_twoval.h"

using namespace amwg;

static std::vector<uint32_t> tables(const std::vector<uint8_t> &x) {   // same recipe as amwg_core.hip two_valued_tables
  const int N = (int)x.size();
  const size_t W = two_valued_words(N);
  std::vector<uint32_t> tab(6 * W, 0u);
  for (int i = 0; i < N; ++i) if (x[i]) tab[(size_t)i >> 5] |= 1u << (i & 31);
  for (size_t k = 1; k < W; ++k) tab[W + k] = tab[W + k - 1] + (uint32_t)__builtin_popcount(tab[k - 1]);
  for (int sym = 1; sym >= 0; --sym) {
    uint32_t *om = tab.data() + (sym ? 2 : 4) * W, *po = om + W;
    int run = 0;
    for (int i = 0; i < N; ++i) { const int v = x[i] ? 1 : 0; if (v == sym) { if (run & 1) om[(size_t)i >> 5] |= 1u << (i & 31); run = 0; } else ++run; }
    for (size_t k = 1; k < W; ++k) po[k] = po[k - 1] + (uint32_t)__builtin_popcount(om[k - 1]);
  }
  return tab;
}

int main(int argc, char **argv) {
  const long cases = argc > 1 ? atol(argv[1]) : 20000;
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  long bad = 0, ties_forced = 0;
  for (long c = 0; c < cases; ++c) {
    const int N = (c % 7 == 0) ? (int)(rng() % 20000) : (int)(rng() % 700);
    const double p_one = (c % 5 == 0) ? 0.02 : (c % 5 == 1 ? 0.97 : U(rng));
    std::vector<uint8_t> x((size_t)N);
    for (auto &v : x) v = U(rng) < p_one;
    const std::vector<uint32_t> tab = tables(x);
    const size_t W = two_valued_words(N);
    const BitData B{tab.data(), tab.data() + W, tab.data() + 2 * W, tab.data() + 3 * W, tab.data() + 4 * W, tab.data() + 5 * W, N};
    double l1 = -std::exp(U(rng) * 15 - 12), l0 = -std::exp(U(rng) * 15 - 12);
    // significands ending in z zero bits tie in the binade 2^(z+1) above the addend
    auto trail = [&](double v, int z, bool setbit) { uint64_t u; memcpy(&u, &v, 8); u = (u >> z) << z; if (setbit) u |= 1ull << z; memcpy(&v, &u, 8); return v; };
    if (c % 3 != 0) { l1 = trail(l1, (int)(rng() % 32), rng() & 1); l0 = trail(l0, (int)(rng() % 32), rng() & 1); ++ties_forced; }
    if (c % 97 == 0) l1 = -1.0;
    if (c % 89 == 0) l0 = -0.5;
    if (c % 1013 == 0) l1 = -INFINITY;
    if (c % 1019 == 0) l0 = NAN;
    if (c % 1021 == 0) l1 = 0.25;
    double acc0 = (c % 2) ? -std::exp(U(rng) * 30 - 5) : (U(rng) - 0.5) * 6;
    if (c % 211 == 0) acc0 = 0.0;
    const double got = two_valued_sum(acc0, l1, l0, B);
    double want = acc0;
    for (int i = 0; i < N; ++i) want = want + (x[i] ? l1 : l0);
    if (memcmp(&got, &want, 8) != 0 && !(got != got && want != want)) {
      if (bad < 5) printf("MISMATCH case %ld N=%d acc0=%a l1=%a l0=%a got=%a want=%a\n", c, N, acc0, l1, l0, got, want);
      ++bad;
    }
  }
  printf("cases=%ld forced_trailing_zero_addends=%ld mismatches=%ld\n", cases, ties_forced, bad);
  return bad ? 1 : 0;
}
