// amwg_eval.h -- device evaluation of the arithmetic building blocks (tests only; amwg_device_eval).
#pragma once
#include "amwg_kernel.h"
#include "amwg_user.h"

namespace amwg {

// ---- device evaluation of the arithmetic building blocks (tests only; amwg_device_eval)
// every scalar density / helper by the ids of oracle/gen_ld_golden.js (tests/golden/ld_values.bin)
AMWG_HD double ld_by_id(int id, double x, double a, double b, double c) {
  switch (id) {
    case 0: return ld_norm(x, a, b);
    case 1: return ld_unif(x, a, b);
    case 2: return ld_beta(x, a, b);
    case 3: return ld_bern(x, a);
    case 4: return ld_pois(x, a);
    case 5: return ld_cauchy(x, a, b);
    case 6: return ld_laplace(x, a, b);
    case 7: return ld_gamma(x, a, b);
    case 8: return ld_invgamma(x, a, b);
    case 9: return ld_lnorm(x, a, b);
    case 10: return ld_pareto(x, a, b);
    case 11: return ld_t(x, a, b, c);
    case 12: return ld_weibull(x, a, b);
    case 13: return ld_logis(x, a, b);
    case 14: return ld_exp(x, a);
    case 15: return ld_binom(x, a, b);
    case 16: return ld_nbinom(x, a, b);
    case 17: return ld_hyper(x, a, b, c);
    case 18: return lgamma_js(x);
    case 19: return lfactorial_js(x);
    case 20: return lchoose_js(x, a);
    case 21: return lbeta_js(x, a);
  }
  return __builtin_nan("");
}

// the one-argument Math.* twins by the ids of amwg_math1 (include/amwg.h)
AMWG_HD double math1_by_id(int fn, double x) {
  switch (fn) {
    case 0: return tanh_v8(x);
    case 1: return atan_v8(x);
    case 2: return log10_v8(x);
    case 3: return sin_v8(x);
    case 4: return cos_v8(x);
    case 5: return tan_v8(x);
    case 6: return asin_v8(x);
    case 7: return acos_v8(x);
    case 8: return sinh_v8(x);
    case 9: return cosh_v8(x);
    case 10: return asinh_v8(x);
    case 11: return acosh_v8(x);
    case 12: return atanh_v8(x);
    case 13: return cbrt_v8(x);
    case 14: return log2_v8(x);
  }
  return __builtin_nan("");
}

__global__ void amwg_ld_eval_kernel(int64_t n, const double *rec /* [n][5]: id, x, a, b, c */, double *out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = ld_by_id((int)rec[i * 5], rec[i * 5 + 1], rec[i * 5 + 2], rec[i * 5 + 3], rec[i * 5 + 4]);
}

__global__ void amwg_eval_kernel(int op, int64_t n, const double *a, const double *b, const double *c, double *out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = a[i], y = b ? b[i] : 0.0, z = c ? c[i] : 0.0;
  double r = 0;
  switch (op) {
    case 0: r = exp_v8(x); break;
    case 1: r = log_v8(x); break;
    case 2: r = __builtin_sqrt(x); break;
    case 3: r = lgamma_js(x); break;
    case 4: r = div_by_invariant(x, y, make_reciprocal(y)); break;
    case 5: r = x / y; break;
    case 6: r = ld_norm(x, y, z); break;
    case 7: r = js_round(x); break;
    case 8: { ChainStream s; s.init((uint64_t)x, (uint64_t)y, (uint64_t)z); r = s.next(); } break;
    case 9: r = ld_pois(x, y); break;
    case 10: r = ld_beta(x, y, z); break;
    case 11: r = ld_bern(x, y); break;
    case 12: r = ld_unif(x, y, z); break;
    case 13: r = pow_v8(x, y); break;
    case 14: r = log1p_v8(x); break;
    case 15: r = expm1_v8(x); break;
    case 16: r = tanh_v8(x); break;
    case 17: r = atan_v8(x); break;
    case 18: r = log10_v8(x); break;
    case 19: r = quot_plain(x, y); break;
    case 20: r = math1_by_id((int)y, x); break;      // y = function id
    case 21: r = atan2_v8(x, y); break;
    case 22: r = hypot3_v8(x, y, z); break;
    case 23: r = hypot2_v8(x, y); break;
    case 24: r = js_mod(x, y); break;                // the `%` of translated closures
    case 25: r = (double)js_toint32(x); break;       // `x | 0`
    case 26: { double lam; r = exp_log_v8(x, lam, exp_log_regs()); } break;          // the fused pair of the Poisson pass: log(exp(x)) ...
    case 27: { double lam; (void)exp_log_v8(x, lam, exp_log_regs()); r = lam; } break;   // ... and its exp(x)
    case 28: r = exp_v8_full(x); break;              // the full fdlibm control flow, for comparison
    case 29: r = log_v8_full(x); break;
    case 30: r = log_v8_full(exp_v8_full(x)); break;
    case 31: { double lam; r = exp_log_v8(x, lam, ExpLogLiterals{}); } break;
    case 32: r = log1p_exp_v8(x); break;                      // softplus in one straight line ...
    case 33: r = log1p_exp_v8(x, exp_log_regs()); break;
    case 34: r = log1p_v8(exp_v8_full(x)); break;             // ... and the full fdlibm control flow it replaces
    case 35: { bool rare = false; r = log1p_exp_v8_open(rare, x); if (rare) r = log1p_exp_cold(x); } break;   // the branch-free form of unrolled loops
  }
  out[i] = r;
}

}  // namespace amwg
