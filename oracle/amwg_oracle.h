/*
 * amwg_oracle.h -- TEST INFRASTRUCTURE ONLY.  See amwg_oracle.c.
 * Nothing under bayes.js_amd/ may include, link or dlopen this.
 */
#ifndef AMWG_ORACLE_H
#define AMWG_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_MODEL_NORMAL = 1, ORC_MODEL_BETA_BERN = 2, ORC_MODEL_HIER_NORMAL = 3, ORC_MODEL_POIS_GLM = 4,
       ORC_MODEL_CALLBACK = 5 /* log_post supplied by the test (a user closure), see orc_set_callback */ };
enum { ORC_REAL = 0, ORC_INT = 1, ORC_BINARY = 2 };
/* log_post(state[P]) summed in the order of `lanes` lanes per chain */
typedef double (*orc_log_post_fn)(const double *state, int lanes, void *ctx);

/* One named parameter of the reference's `params` object after
 * complete_params() (mcmc.js:357-403), flattened row-major. */
typedef struct {
  int32_t type;      /* ORC_REAL | ORC_INT (Metropolis, mcmc.js:517-553) | ORC_BINARY (BinaryStepper, mcmc.js:753-767) */
  int32_t len;       /* prod(dim) */
  int32_t top;       /* dim[0]; the only dimension that is shuffled (mcmc.js:244-258) */
  int32_t multidim;  /* 0 iff dim equals [1] (dispatch rule of mcmc.js:846-857) */
  double lower, upper;
} orc_param;

/* Per scalar component stepper options (mcmc.js:500-505), already merged. */
typedef struct {
  double prop_log_scale, max_adaptation, initial_adaptation, target_accept_rate;
  double batch_size;       /* a JS number, compared / divided as such (mcmc.js:538, 543) */
  int32_t is_adapting;
} orc_comp_opt;

typedef struct {
  int32_t model;
  int32_t n_obs;
  const double *x;   /* normal: x[N]; beta_bern: x[N] in {0,1}; hier_normal: y[N]; pois_glm: X[N][K] row-major */
  const double *y;   /* pois_glm: counts y[N] */
  const int32_t *g;  /* hier_normal: group of obs i */
  int32_t G, K;
  double hyper[8];   /* prior hyper-parameters, same meaning as amwg_model_desc.hyper */
} orc_data;

typedef struct orc_chain orc_chain;

/* lanes = summation order of the observation loop: 1 = the reference's
 * sequential `lp += term` order; L>1 = L strided partial sums + xor butterfly
 * (offsets 1,2,4..), which is the order the HIP kernel uses with L lanes per chain. */
/* ORC_MODEL_CALLBACK: the function orc_create and every step will call; set it BEFORE orc_create */
void orc_set_callback(orc_log_post_fn fn, void *ctx);
orc_chain *orc_create(const orc_data *d, const orc_param *params, int n_params, const double *init /*P*/,
                      const orc_comp_opt *opts /*P*/, uint64_t seed, uint64_t chain, int lanes);
/* group-local evaluation of the hierarchical family (see amwg_oracle.c, gl_*): 0 on success, -1 if the model / data / lane count does not
 * meet its preconditions.  Call right after orc_create (it re-forms lp_curr in the mode's own summation order). */
int orc_set_group_local(orc_chain *c, int flag);
void orc_destroy(orc_chain *c);
int orc_num_components(const orc_chain *c);
void orc_burn(orc_chain *c, int64_t n);
/* draws: [ceil(n/thin)][P], the state BEFORE step i for every i % thin == 0 (mcmc.js:1020-1027) */
void orc_sample(orc_chain *c, int64_t n, int64_t thin, double *draws);
void orc_set_adapting(orc_chain *c, int flag);
void orc_get_state(const orc_chain *c, double *state /*P*/);
void orc_get_info(const orc_chain *c, double *prop_log_scale, int32_t *acceptance_count, int32_t *iterations_since_adaption,
                  int32_t *batch_count, int64_t *accepts, int64_t *inbounds /* each P, may be NULL */);
uint64_t orc_uniforms_used(const orc_chain *c);
double orc_log_post(orc_chain *c);
double orc_log_post_unhoisted(orc_chain *c); /* same value, every ld.* call spelled out per observation */
void orc_named_order(const orc_chain *c, int32_t *order /*n_params*/);

/* exposed pieces, pinned individually by tests */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double orc_uniform(uint64_t seed, uint64_t chain, uint64_t index);
double orc_exp(double x);
double orc_log(double x);
double orc_js_round(double x);
double orc_ld_norm(double x, double mean, double sd);
double orc_ld_unif(double x, double lo, double hi);
double orc_ld_beta(double x, double a, double b);
double orc_ld_bern(double x, double p);
double orc_ld_pois(double x, double lambda);
double orc_lgamma(double x);
double orc_pow(double x, double y);   /* V8 Math.pow */
double orc_log1p(double x);          /* V8 Math.log1p */
double orc_expm1(double x);          /* V8 Math.expm1 */
double orc_tanh(double x);           /* V8 Math.tanh */
double orc_atan(double x);           /* V8 Math.atan */
double orc_log10(double x);          /* V8 Math.log10 */
/* every scalar density / helper of distributions.js by id (oracle/gen_ld_golden.js lists the ids) */
double orc_ld(int id, double x, double a, double b, double c);

#ifdef __cplusplus
}
#endif
#endif
