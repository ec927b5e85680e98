/* amwg_selftest.h -- TEST BUILD ONLY (libamwg_selftest.so = libamwg.so's sources compiled with -DAMWG_SELFTEST): the arithmetic building
 * blocks of the kernel exported one by one, so that each can be pinned against V8 / the reference's distributions.js on the host and on
 * the device.  None of this is in the product library. */
#ifndef AMWG_SELFTEST_H
#define AMWG_SELFTEST_H
#include "amwg.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Device evaluation: op in {0:exp,1:log,2:sqrt,3:lgamma,4:a/b via hoisted reciprocal,5:ld_norm(a,b,c),...};
 * a,b,c host arrays of n doubles (b,c may be NULL), out host array of n doubles. */
int amwg_device_eval(int32_t device, int32_t op, int64_t n, const double *a, const double *b, const double *c, double *out);
double amwg_pow(double x, double y);   /* bit-identical to V8 Math.pow */
double amwg_log1p(double x);           /* bit-identical to V8 Math.log1p */
double amwg_expm1(double x);           /* bit-identical to V8 Math.expm1 */
double amwg_math1(int32_t fn, double x); /* fn 0 tanh, 1 atan, 2 log10, 3 sin, 4 cos, 5 tan, 6 asin, 7 acos, 8 sinh, 9 cosh, 10 asinh, 11 acosh, 12 atanh, 13 cbrt, 14 log2:
                                            bit-identical to V8's Math.* (host build of the kernel source, csrc/amwg_math.h + amwg_trig.h) */
double amwg_math2(int32_t fn, double x, double y); /* fn 0: Math.atan2(x, y), 1: Math.hypot(x, y), 2: x % y (JavaScript), 3: x | 0 (ToInt32; y ignored) */
double amwg_hypot3(double x, double y, double z);   /* Math.hypot(x, y, z) */
/* Every scalar ld.* density and helper of distributions.js by id (0 norm 1 unif 2 beta 3 bern 4 pois 5 cauchy
 * 6 laplace 7 gamma 8 invgamma 9 lnorm 10 pareto 11 t 12 weibull 13 logis 14 exp 15 binom 16 nbinom 17 hyper
 * 18 lgamma 19 lfactorial 20 lchoose 21 lbeta): host evaluation of the kernel's own source, and the same on
 * the device for n records of {id, x, a, b, c}. */
double amwg_ld_host(int32_t id, double x, double a, double b, double c);
int amwg_ld_device(int32_t device, int64_t n, const double *records, double *out);
/* out_fast_forward[j] = the two-valued sequential sum of csrc/amwg_twoval.h (exact fast-forward over binades) and
 * out_term_by_term[j] = the plain loop, both on the device, over the n observations x (0/1) from acc0[j] with addends
 * l1[j] (x_i = 1) and l0[j] (x_i = 0); j < m. */
int amwg_two_valued_sum_check(int32_t device, const double *x, int32_t n, int64_t m, const double *acc0, const double *l1, const double *l0,
                              double *out_fast_forward, double *out_term_by_term);

/* The host machinery behind sample()'s copy-out (csrc/amwg_core.hip Prefaulter: huge pages, MADV_POPULATE_WRITE, helper threads that touch the destination ahead of
 * the device-to-host copies) on a caller's buffer, cut into n_chunks chunks, with `threads` helpers (0 = the calling thread alone).  Changes no byte.  No GPU. */
int amwg_prefault_selftest(char *buf, size_t bytes, int32_t n_chunks, int32_t threads);

/* BOUND AUDIT build only (libamwg_audit.so = the library's sources compiled with -DAMWG_AUDIT; tools/bound_audit.py, tests/test_gpu_bound_audit.py).  The kernels that
 * decide accept tests from a cheaper value A of log_post and a bound eps (csrc/amwg_kernel.h "certified decisions") evaluate the reference's expression E in EVERY
 * update there as well and record how far apart the two really are:
 *   per_chain [4][chains]: max |A - E| / eps,  max |dA - dE| / eta,  audited decisions,  certified verdicts that contradict exp(dE) > u (must be 0)
 *   hist      [2][64]:     counts of the two ratios by binary exponent, bin b >= 1 holding [2^(b-40), 2^(b-39)) -- bins >= 40 are violated bounds; bin 0: exactly equal
 * reset != 0 clears both afterwards.  amwg_options::test_bound_shift may be negative (down to -60) in this build.  Not in the product library. */
#if defined(AMWG_AUDIT) || defined(AMWG_X_PHASES)
int amwg_audit_fetch(amwg_sampler *s, double *per_chain, uint64_t *hist, int32_t reset);
#endif

#ifdef __cplusplus
}
#endif
#endif
