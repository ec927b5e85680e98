// div_fuzz.cpp -- TEST INFRASTRUCTURE: csrc/amwg_div.h (correctly rounded division by a loop-invariant divisor via a
// double-double reciprocal) against IEEE '/', on the host, inside the range the kernels use it in (divisor 2^-200..2^200,
// numerator 2^-600..2^600 or zero), incl. all-ones / power-of-two significands.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "amwg_div.h"

using namespace amwg;

int main(int argc, char **argv) {
  const long cases = argc > 1 ? atol(argv[1]) : 1000000;
  std::mt19937_64 rng(777);
  auto rnd = [&](int elo, int ehi, int kind) {
    uint64_t mant = rng() & 0x000fffffffffffffull;
    if (kind == 1) mant = 0x000fffffffffffffull;
    if (kind == 2) mant = 0;
    if (kind == 3) mant = 0x000fffffffffffffull & ~(rng() & 0xffull);
    const uint64_t e = (uint64_t)(1023 + elo + (int)(rng() % (uint64_t)(ehi - elo + 1)));
    const uint64_t u = (e << 52) | mant;
    double v; memcpy(&v, &u, 8); return v;
  };
  long bad = 0;
  for (long c = 0; c < cases; ++c) {
    const double b = rnd(-200, 200, (int)(c % 11 == 0 ? 1 : (c % 13 == 0 ? 2 : (c % 17 == 0 ? 3 : 0))));
    const Reciprocal y = make_reciprocal(b);
    for (int k = 0; k < 4; ++k) {
      const double a = (c % 101 == 0 && k == 0) ? 0.0 : rnd(-600, 600, (int)((c + k) % 7 == 0 ? 1 : ((c + k) % 5 == 0 ? 2 : 0)));
      const double q = div_by_invariant(a, b, y), w = a / b;
      if (memcmp(&q, &w, 8) != 0) { if (bad < 5) printf("MISMATCH a=%a b=%a got=%a want=%a\n", a, b, q, w); ++bad; }
    }
  }
  printf("divisors=%ld quotients=%ld mismatches=%ld\n", cases, cases * 4, bad);
  return bad ? 1 : 0;
}
