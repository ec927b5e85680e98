// explog_fuzz.cpp -- TEST INFRASTRUCTURE: the straight-line exp_v8 and the fused exp_log_v8 of csrc/amwg_math.h against the full fdlibm
// control flow (exp_v8_full, log_v8_full, themselves pinned against Node's Math.exp / Math.log by tests/test_core_host.py), on the host:
// random arguments over the whole range, every high word next to the thresholds the straight-line forms replace (0.5 ln2, 1.5 ln2, 2^-28,
// 708, 1.0), and arguments whose exp() lands next to the significand thresholds of log (0x6a09c, 0x6147a, 0x6b851, |f| < 2^-20).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "amwg_math.h"

using namespace amwg;

static long bad = 0, seen = 0, fused_common = 0, fused_other = 0, form_a = 0;
static void check(double x) {
  const double we = exp_v8_full(x), wl = log_v8_full(we);
  const double ge = exp_v8(x);
  double lam_l, lam_r;
  const double gl = exp_log_v8(x, lam_l, ExpLogLiterals{});
  const ExpLogRegs regs = exp_log_regs();
  const double gr = exp_log_v8(x, lam_r, regs);
  ++seen;
  {   // the two-wide branch-free form: a fresh argument beside the previous one, rare lanes patched as its caller does
    static double prev = 0.5;
    const double xs[2] = {x, prev};
    double lam2[2], lg2[2];
    bool rare2[2];
    exp_log_v8_open<2>(xs, lam2, lg2, rare2, regs);
    for (int u = 0; u < 2; ++u)
      if (rare2[u]) { lam2[u] = exp_v8(xs[u]); lg2[u] = log_v8(lam2[u]); }
    const double we1 = exp_v8_full(prev), wl1 = log_v8_full(we1);
    auto same2 = [](double a, double b) { return memcmp(&a, &b, 8) == 0 || (a != a && b != b); };
    if (!same2(lam2[0], we) || !same2(lg2[0], wl) || !same2(lam2[1], we1) || !same2(lg2[1], wl1)) {
      if (bad < 10) printf("MISMATCH (two-wide) x=%a prev=%a\n", x, prev);
      ++bad;
    }
    prev = x;
  }
  auto same = [](double a, double b) { return memcmp(&a, &b, 8) == 0 || (a != a && b != b); };
  if (!same(ge, we) || !same(lam_l, we) || !same(lam_r, we) || !same(gl, wl) || !same(gr, wl)) {
    if (bad < 10) printf("MISMATCH x=%a exp=%a want=%a | log=%a want=%a\n", x, ge, we, gl, wl);
    ++bad;
  }
  if (!exp_is_rare(x)) {
    const ExpParts e = exp_parts(x, ExpLogLiterals{});
    const uint32_t tmp = (uint32_t)hi_word(e.y) - 0x3fe6a09cu;
    if (tmp < 0x100000u) ++fused_common; else ++fused_other;
    if (tmp < 0x100000u && !((tmp - (0x3fe6b852u - 0x3fe6a09cu)) <= (0x3ff61479u - 0x3fe6b852u))) ++form_a;
  }
}

int main(int argc, char **argv) {
  const long cases = argc > 1 ? atol(argv[1]) : 1000000;
  std::mt19937_64 rng(4242);
  auto bits = [](uint64_t u) { double v; memcpy(&v, &u, 8); return v; };
  // 1. random arguments: uniform in the range a log link produces, and log-uniform magnitudes of both signs
  std::uniform_real_distribution<double> U(-30.0, 30.0), W(-745.5, 710.0);
  for (long c = 0; c < cases; ++c) {
    check(U(rng));
    check(W(rng));
    const uint64_t e = 1023 - 60 + rng() % 71;          // 2^-60 .. 2^10
    check(bits(((rng() & 1) << 63) | (e << 52) | (rng() & 0x000fffffffffffffull)));
  }
  // 2. high words next to every threshold of the argument reduction, both signs, random / extreme low words
  const uint32_t hw[] = {0x3fd62e42u, 0x3ff0a2b2u, 0x3e300000u, 0x40862000u, 0x40862e42u, 0x3ff00000u, 0x40874910u, 0x3fe62e42u, 0x3ff62e42u};
  for (uint32_t h : hw)
    for (int d = -3; d <= 3; ++d)
      for (int sgn = 0; sgn < 2; ++sgn)
        for (long c = 0; c < cases / 50 + 8; ++c) {
          uint32_t lo = (uint32_t)rng();
          if (c == 0) lo = 0; if (c == 1) lo = 0xffffffffu; if (c == 2) lo = 0xfefa39efu; if (c == 3) lo = 0xfefa39eeu; if (c == 4) lo = 0xfefa39f0u;
          if (c == 5) lo = 0x3f3bab73u; if (c == 6) lo = 0x3f3bab72u; if (c == 7) lo = 0x3f3bab74u;
          check(bits(((uint64_t)sgn << 63) | ((uint64_t)(h + d) << 32) | lo));
        }
  // 3. arguments k ln2 + log(m) with m next to the significand thresholds of log: exp() then lands on either side of them
  const uint32_t mw[] = {0x3ff6a09cu, 0x3fe6a09cu, 0x3ff6147au, 0x3fe6b851u, 0x3ff6b851u, 0x3fe6147au, 0x3ff00000u, 0x3feffffeu, 0x3ff6a09eu, 0x3fe6a09eu};
  for (uint32_t h : mw)
    for (int d = -4; d <= 4; ++d)
      for (long c = 0; c < cases / 20 + 4; ++c) {
        const double m = bits(((uint64_t)(h + d) << 32) | (uint32_t)rng());
        const int k = (int)(rng() % 61) - 30;
        const double x = (double)k * 0.6931471805599453 + std::log(m);
        check(x);
        check(std::nextafter(x, 1e300));
        check(std::nextafter(x, -1e300));
      }
  // 4. specials
  const double sp[] = {0.0, -0.0, 1.0, -1.0, INFINITY, -INFINITY, NAN, 709.782712893384, 709.7827128933841, -745.1332191019411, -745.1332191019412,
                       -708.0, 708.0, 1e-300, -1e-300, 5e-324, 0.34657359027997264, -0.34657359027997264, 1.0397207708399179, -1.0397207708399179};
  for (double x : sp) check(x);
  // 5. exp_bounded (the certified pass of the Poisson family): NOT V8's exp -- its relative distance from exp, measured against long double, must stay
  // below the 2^-46 its users add to their bounds (kExpBoundedRel); literal and register-struct constants give the same bits
  {
    long double worst = 0;
    const ExpTaylorRegs tr = exp_taylor_regs();
    std::uniform_real_distribution<double> V(-700.0, 700.0), S(-2.0, 2.0);
    auto one = [&](double x) {
      const double g = exp_bounded(x, ExpTaylorLiterals{}), g2 = exp_bounded(x, tr);
      const long double w = expl((long double)x);
      const long double rel = fabsl(((long double)g - w) / w);
      if (rel > worst) worst = rel;
      if (memcmp(&g, &g2, 8) != 0) { if (bad < 10) printf("MISMATCH (exp_bounded literals vs registers) x=%a\n", x); ++bad; }
    };
    for (long c = 0; c < cases; ++c) { one(V(rng)); one(S(rng)); one((double)((long)(rng() % 2001) - 1000) * 0.6931471805599453 * 0.5 + S(rng) * 1e-9); }
    for (double x : {0.0, -0.0, 700.0, -700.0, 0.34657359027997264, -0.34657359027997264, 1e-300, -1e-300}) one(x);
    printf("exp_bounded_worst_rel=%.3Le (2^-46 = %.3e)\n", worst, 0x1p-46);
    if (!(worst < 0x1p-48L)) { printf("exp_bounded is not within a quarter of its stated bound\n"); ++bad; }
  }
  printf("arguments=%ld fused_common=%ld (form a: %ld) other_split=%ld mismatches=%ld\n", seen, fused_common, form_a, fused_other, bad);
  if (fused_other < 1000 || form_a < 1000) { printf("coverage too thin\n"); return 2; }
  return bad ? 1 : 0;
}
