// norm_inv_fold.cpp -- TEST INFRASTRUCTURE (host): translate.js folds norm_inv(<literal>) into literals at translation time (foldConstantNormInv); every folded
// constant must be the bits csrc/amwg_user.h norm_inv() computes.  stdin: lines "<sd as hex float> NormInv{c, den, Reciprocal{hi, lo}, true} /* ... */"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "amwg_user.h"
using namespace amwg;
int main() {
  char line[1024];
  long n = 0, bad = 0, unfolded = 0;
  while (fgets(line, sizeof line, stdin)) {
    char *sp = strchr(line, ' ');
    if (!sp) continue;
    *sp = 0;
    const double sd = strtod(line, nullptr);
    const char *rest = sp + 1;
    if (strncmp(rest, "NormInv{", 8) != 0) { ++unfolded; continue; }
    double c, den, hi, lo;
    const char *p = rest + 8;
    char *e;
    c = strtod(p, &e); p = e + 2; den = strtod(p, &e); p = strstr(e, "Reciprocal{") + 11; hi = strtod(p, &e); p = e + 2; lo = strtod(p, &e);
    const NormInv k = norm_inv(sd);
    ++n;
    if (memcmp(&c, &k.c, 8) || memcmp(&den, &k.den, 8) || memcmp(&hi, &k.y.hi, 8) || memcmp(&lo, &k.y.lo, 8) || !k.fast) {
      if (bad < 10) printf("MISMATCH sd=%a: folded {%a, %a, %a, %a} device {%a, %a, %a, %a, fast %d}\n", sd, c, den, hi, lo, k.c, k.den, k.y.hi, k.y.lo, (int)k.fast);
      ++bad;
    }
  }
  printf("checked=%ld unfolded=%ld mismatches=%ld\n", n, unfolded, bad);
  return bad ? 1 : 0;
}
