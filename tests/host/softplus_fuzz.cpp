// softplus_fuzz.cpp -- TEST INFRASTRUCTURE: the straight-line log1p_exp_v8 of csrc/amwg_math.h (softplus: the logistic log-likelihood's
// Math.log1p(Math.exp(eta)) in one pass of selects) against log1p_v8(exp_v8_full(x)) -- fdlibm's full control flow, both pinned against
// Node's Math.log1p / Math.exp by tests/test_core_host.py -- on the host: random arguments, arguments whose exp() lands next to every
// threshold the selects replace (sqrt(2)-1; 1 + exp(x) next to a power of two and next to sqrt(2) 2^k; 2^-29; 2^53), the edges of the
// straight line's own range (-20, 36) and exp's rare arguments.  Also counts that every form was visited.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "amwg_math.h"

using namespace amwg;

static long n_open_clean = 0, bad = 0, seen = 0, n_small = 0, n_up = 0, n_down = 0, n_cold = 0, n_k0 = 0;
static void check(double x) {
  const double want = log1p_v8(exp_v8_full(x));
  const double got = log1p_exp_v8(x), got_r = log1p_exp_v8(x, exp_log_regs());
  ++seen;
  auto same = [](double a, double b) { return memcmp(&a, &b, 8) == 0 || (a != a && b != b); };
  {   // the branch-free form as translate.js uses it: four arguments back to back, one flag, flagged lanes through the full functions
    static double prev[3] = {0.5, -3.0, 7.0};
    const double xs[4] = {x, prev[0], prev[1], prev[2]};
    double t[4];
    bool rare = false;
    for (int u = 0; u < 4; ++u) t[u] = log1p_exp_v8_open(rare, xs[u]);
    if (rare) for (int u = 0; u < 4; ++u) t[u] = log1p_exp_cold(xs[u]);
    else ++n_open_clean;
    for (int u = 0; u < 4; ++u)
      if (!same(t[u], log1p_v8(exp_v8_full(xs[u])))) { if (bad < 10) printf("MISMATCH (open) x=%a\n", xs[u]); ++bad; }
    prev[2] = prev[1]; prev[1] = prev[0]; prev[0] = x;
  }
  if (!same(got, want) || !same(got_r, want)) {
    if (bad < 10) printf("MISMATCH x=%a got=%a want=%a\n", x, got, want);
    ++bad;
  }
  if (!(x >= -20.0 && x <= 36.0) || exp_is_rare(x)) { ++n_cold; return; }
  const double v = exp_v8_full(x);
  if (hi_word(v) < 0x3FDA827A) { ++n_small; return; }
  const double u = 1.0 + v;
  const int32_t mant = hi_word(u) & 0xfffff;
  if ((hi_word(u) >> 20) - 1023 == 0) ++n_k0;
  if (mant >= 0x6a09e) ++n_up; else ++n_down;
}

int main(int argc, char **argv) {
  const long cases = argc > 1 ? atol(argv[1]) : 1000000;
  std::mt19937_64 rng(777);
  auto bits = [](uint64_t u) { double v; memcpy(&v, &u, 8); return v; };
  std::uniform_real_distribution<double> U(-8.0, 8.0), V(-22.0, 38.0), W(-745.5, 710.0);
  for (long c = 0; c < cases; ++c) {
    check(U(rng));
    check(V(rng));
    if (c % 4 == 0) check(W(rng));
    const uint64_t e = 1023 - 60 + rng() % 67;          // 2^-60 .. 2^6
    check(bits(((rng() & 1) << 63) | (e << 52) | (rng() & 0x000fffffffffffffull)));
  }
  // exp(x) next to sqrt(2) - 1 (0x3FDA827A), 2^-29, 2^53 and a few plain values
  const uint32_t vw[] = {0x3FDA827Au, 0x3e200000u, 0x43400000u, 0x3ff00000u, 0x3fe00000u, 0x40000000u};
  for (uint32_t h : vw)
    for (int d = -3; d <= 3; ++d)
      for (long c = 0; c < cases / 20 + 8; ++c) {
        uint32_t lo = (uint32_t)rng();
        if (c == 0) lo = 0; if (c == 1) lo = 0xffffffffu;
        const double x = std::log(bits(((uint64_t)(h + d) << 32) | lo));
        check(x); check(std::nextafter(x, 1e300)); check(std::nextafter(x, -1e300));
      }
  // 1 + exp(x) = m 2^k with m next to sqrt(2) (the choice of the half) and next to 1 / 2 (|f| < 2^-20), k = 0 .. 52
  const uint32_t mw[] = {0x3ff6a09eu, 0x3ff00000u, 0x3ffffffdu, 0x3ff00004u, 0x3ff80000u};
  for (uint32_t h : mw)
    for (int d = -3; d <= 3; ++d)
      for (long c = 0; c < cases / 10 + 8; ++c) {
        const double m = bits(((uint64_t)(h + d) << 32) | (uint32_t)rng());
        const int k = (int)(rng() % 53);
        const double t = std::ldexp(m, k) - 1.0;
        if (!(t > 0)) continue;
        const double x = std::log(t);
        check(x); check(std::nextafter(x, 1e300)); check(std::nextafter(x, -1e300));
      }
  // the edges of the straight line's range, exp's rare arguments, specials
  const double ed[] = {-20.0, 36.0, -20.10126823623841, 36.7368005696771, 30.0, 0.0, -0.0, 1.0, -1.0, INFINITY, -INFINITY, NAN, 709.782712893384,
                       709.7827128933841, -745.1332191019411, -745.1332191019412, -708.0, 708.0, 1e-300, -1e-300, 5e-324, 0.34657359027997264,
                       -0.34657359027997264, 1.0397207708399179, -1.0397207708399179, -0.8813735870195429, 0.8813735870195429};
  for (double x : ed)
    for (int s = -40; s <= 40; ++s) {
      double y = x;
      for (int j = 0; j < (s < 0 ? -s : s); ++j) y = std::nextafter(y, s < 0 ? -1e300 : 1e300);
      check(y);
    }
  // Node's own values (tests/golden/v8_softplus_pairs.bin, oracle/gen_softplus_pairs.js)
  long v8_pairs = 0, v8_bad = 0;
  if (argc > 2) {
    if (FILE *fp = fopen(argv[2], "rb")) {
      double rec[2];
      while (fread(rec, 8, 2, fp) == 2) {
        const double got = log1p_exp_v8(rec[0]);
        ++v8_pairs;
        if (!(memcmp(&got, &rec[1], 8) == 0 || (got != got && rec[1] != rec[1]))) { if (v8_bad < 10) printf("V8 MISMATCH x=%a got=%a want=%a\n", rec[0], got, rec[1]); ++v8_bad; }
      }
      fclose(fp);
    }
    printf("v8_pairs=%ld v8_mismatches=%ld\n", v8_pairs, v8_bad);
    if (v8_pairs < 1000 || v8_bad) return 3;
  }
  printf("open_form_unflagged_quads=%ld\n", n_open_clean);
  if (n_open_clean < 1000) { printf("coverage too thin (open form)\n"); return 2; }
  printf("arguments=%ld no_reduction=%ld reduced_low_half=%ld (k = 0 before: %ld) reduced_high_half=%ld cold=%ld mismatches=%ld\n", seen, n_small, n_down, n_k0, n_up, n_cold, bad);
  if (n_small < 1000 || n_down < 1000 || n_up < 1000 || n_cold < 1000 || n_k0 < 1000) { printf("coverage too thin\n"); return 2; }
  return bad ? 1 : 0;
}
