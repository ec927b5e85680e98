/*
 * amwg.h -- C ABI of libamwg.so: many-chain Adaptive-Metropolis-within-Gibbs on MI355X.
 *
 * This is the drop-in boundary for ONE path of rasmusab/bayes.js:
 *
 *     new mcmc.AmwgSampler(params, log_post, data, options)      mcmc.js:1090-1099 (Sampler ctor :940-966)
 *     sampler.burn(n)                                            mcmc.js:1035-1039
 *     sampler.sample(n)                                          mcmc.js:1005-1030
 *     sampler.thin(k) / .monitor(names)                          mcmc.js:1053-1055 / :1045-1047  (host side; see thin arg)
 *     sampler.start_adaptation() / .stop_adaptation()            mcmc.js:1060-1073
 *     sampler.info()                                             mcmc.js:977-980 -> :906-912 -> :563-571
 *     sampler.state                                              mcmc.js:964
 *
 * Each entry point below names the reference method it replaces.  The reference has no
 * FFI of its own (it is two plain-JS files); the binding a maintainer adds is the N-API
 * shim bayes.js_amd/csrc/amwg_napi.c, described in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a
 * negative AMWG_E* code, with text available from amwg_last_error() (thread-local).  The
 * caller owns every host buffer it passes; the library owns all device memory it allocates.
 * Calls on one sampler must be serialised by the caller (the reference is single-threaded).
 * Many independent chains run per sampler; chain c (global id chain_offset + c) uses the
 * Philox4x32-10 stream keyed by (seed, global id), so results do not depend on how chains
 * are sharded over GPUs.
 */
#ifndef AMWG_H
#define AMWG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMWG_OK 0
#define AMWG_EINVAL (-1)   /* bad argument / unsupported model description */
#define AMWG_EHIP (-2)     /* HIP runtime error (no device, launch failure, out of memory) */
#define AMWG_ESIZE (-3)    /* caller buffer too small */

/* Built-in model registry: the user's `log_post(state, data)` closure (mcmc.js:958-960),
 * restricted to the BASELINE.json model families.  Each is the README pattern
 * (README.md:149-164): priors, then a loop over the data adding one ld.* term per observation. */
enum {
  AMWG_MODEL_NORMAL = 1,      /* mu ~ norm(0,100); sigma ~ unif(0,100); x_i ~ norm(mu, sigma)            (README.md:22-36) */
  AMWG_MODEL_BETA_BERN = 2,   /* theta ~ beta(2,2); x_i ~ bern(theta)                                    (README.md:149-164) */
  AMWG_MODEL_HIER_NORMAL = 3, /* mu ~ norm(0,100); sigma ~ unif(0,100); theta_g ~ norm(mu,10); y_i ~ norm(theta[g_i], sigma) */
  AMWG_MODEL_POIS_GLM = 4     /* beta_k ~ norm(0,10); cp ~ unif(0,N-1); y_i ~ pois(exp(X_i.beta[0:K] + [i>=cp] beta[7])) */
};

enum { AMWG_REAL = 0, AMWG_INT = 1, AMWG_BINARY = 2, AMWG_FIXED = 3 };

/* One named parameter AFTER complete_params() (mcmc.js:357-403), flattened row-major.
 * The order of the array is Object.keys(params) order (mcmc.js:839). */
typedef struct {
  int32_t type;      /* AMWG_REAL | AMWG_INT (Metropolis steppers, mcmc.js:517-553) | AMWG_BINARY (BinaryStepper, mcmc.js:753-767) |
                        AMWG_FIXED: a state entry log_post reads but this sampler never steps -- the rest of the shared state
                        object of a stand-alone stepper (mcmc.js:424-431, 1109-1115: every stepper class takes the whole state and
                        moves one parameter of it).  Fixed entries come after all stepped ones; amwg_create_user only. */
  int32_t len;       /* prod(dim) */
  int32_t top;       /* dim[0]: the only dimension whose visiting order is shuffled (mcmc.js:244-258) */
  int32_t multidim;  /* 0 iff dim equals [1]  (stepper dispatch rule, mcmc.js:846-857) */
  double lower, upper;
} amwg_param_desc;

/* Stepper options of one scalar component, already merged the way AmwgStepper merges them
 * (mcmc.js:869-878) and defaulted (mcmc.js:500-505). */
typedef struct {
  double prop_log_scale;      /* default 0    */
  double max_adaptation;      /* default 0.33 */
  double initial_adaptation;  /* default 1.0  */
  double target_accept_rate;  /* default 0.44 */
  double batch_size;          /* default 50; a JS number: compared and divided as such (mcmc.js:538, 543), so 50.5 or 0 behave as in the reference */
  int32_t is_adapting;        /* default 1    */
} amwg_comp_opt;

/* The `data` argument (mcmc.js:942): host arrays, copied to the device by amwg_create. */
typedef struct {
  int32_t model;     /* AMWG_MODEL_* */
  int32_t n_obs;
  const double *x;   /* NORMAL: x[N]; BETA_BERN: x[N]; HIER_NORMAL: y[N]; POIS_GLM: X[N][K] row-major */
  const double *y;   /* POIS_GLM: counts y[N]; otherwise NULL */
  const int32_t *g;  /* HIER_NORMAL: group index of observation i, 0 <= g_i < G; otherwise NULL */
  int32_t G;         /* HIER_NORMAL: number of groups (= len of the first param) */
  int32_t K;         /* POIS_GLM: number of real columns (7) */
  /* Prior hyper-parameters (the numeric literals of the user's closure):
   *   NORMAL      {m0, s0, a, b}        mu ~ norm(m0,s0); sigma ~ unif(a,b)              default {0,100,0,100}
   *   BETA_BERN   {a, b}                theta ~ beta(a,b)                                default {2,2}
   *   HIER_NORMAL {m0, s0, a, b, tau}   mu ~ norm(m0,s0); sigma ~ unif(a,b); theta_g ~ norm(mu,tau)   default {0,100,0,100,10}
   *   POIS_GLM    {m, s}                beta_k ~ norm(m,s); cp ~ unif(0,N-1)             default {0,10} */
  double hyper[8];
} amwg_model_desc;

#define AMWG_LANES_FASTEST (-1)
#define AMWG_LANES_AUTOTUNE (-2)
typedef struct {
  int64_t chains;          /* independent chains on this sampler (>= 1) */
  uint64_t seed;           /* Philox key */
  uint64_t chain_offset;   /* global id of local chain 0 (multi-GPU sharding) */
  int32_t device;          /* HIP device ordinal */
  int32_t lanes_per_chain; /* lanes that split one chain's observation loop: a power of two 1..1024 (above 64 the chain is one workgroup of
                              several wavefronts, each a replica of the scalar logic; few chains, long data loops), or
                              0 = auto, REFERENCE ORDER FIRST: one lane per chain -- the reference's own sequential `lp += term`, every draw of a
                                  seeded run bit-identical to the reference -- whenever the cost model prices it within 12 % of the cheapest
                                  geometry, else the cheapest (decisions identical to the reference, doubles in the G-lane order);
                              AMWG_LANES_FASTEST (-1) = the cheapest geometry regardless of summation order;
                              AMWG_LANES_AUTOTUNE (-2) = measured instead of modelled: at construction every lane count that fits runs a few
                                  steps on the device (the chain state is saved and restored, tuning leaves no trace) and the fastest is kept --
                                  one lane per chain whenever it measures within 12 % of the fastest.  The pick may differ between machines,
                                  and with it the summation order (decisions stay the reference's); amwg_tuning reports the timings */
  int32_t block_threads;   /* 0 = auto; else multiple of 64, <= 1024.  (Hierarchical family at 64 lanes per chain: its sweep kernel runs in workgroups of
                              at most 512 threads; asking for more selects the kernel that evaluates everything, as full_evaluation = 1 does.) */
  int32_t steps_per_launch;/* 0 = auto: one launch per burn call (up to 65535 steps); a sample call that will be fetched is cut into launches of ~32 MB
                            * of recorded rows, so that the rows of one launch are copied out while the next ones run.  Results never depend on it. */
  int32_t exact_division;  /* 0 = default: result-preserving shortcuts (hoisted-reciprocal division, fast-forward of two-valued sums), bit-identical
                              to the plain schedule and tested against it; 1 = the reference's operation schedule: IEEE '/', term-by-term sums */
  int32_t group_local;     /* 0 = default.  1 = GROUP-LOCAL evaluation of the hierarchical family (AMWG_MODEL_HIER_NORMAL, any labels g_i in [0, G), G <= 64
                              -- since round 4: the library deals the 64 lanes of a chain's wavefront to the groups in aligned power-of-two blocks and
                              lays the data out lane-major, csrc/amwg_gl.h; forces 64 lanes per chain; anything else is AMWG_EINVAL): a proposal for theta_g is decided on the
                              difference of ITS group's terms only -- and the G proposals of a sweep are evaluated in ONE pass over the data, every lane
                              with the proposed mean of its own group -- instead of on two full sums (mcmc.js:524-526 evaluates the whole log_post twice
                              per update).  NOT the reference's operation schedule: the doubles follow the order restated in oracle/amwg_oracle.c (gl_*),
                              bit for bit; accept decisions, adaptation and uniforms consumed equal the reference's on every golden and over the
                              1e10-decision campaign of tools/flip_rate.py (a decision can differ only when the accept uniform falls inside the ~1e-12-relative
                              sliver between the two summation orders).  Opt-in, reported separately by bench.py */
  int32_t full_evaluation; /* How accept tests get their log_post values (mcmc.js:524-528).  All three settings give the same draws; they differ in what is evaluated
                              how often, and in the summation order the rare close call is decided in.
                              0 = default: CERTIFIED DECISIONS where the library has them (csrc/amwg_kernel.h; the kernels amwg_step_kernel_cert / amwg_sweep_kernel_cert):
                              the Normal family at one lane per chain, the Poisson family at 16 lanes, the hierarchical family on a wavefront per chain in its row
                              layout (the proposals of a whole sweep over theta drawn ahead -- in stream order: nothing an update draws depends on an earlier decision --
                              and all its accept tests decided at once).  The test is decided from a cheaper value of log_post with a rigorous bound on its distance from
                              the reference's expression summed in the REFERENCE's order (one running sum); a uniform inside the bound (1e-7 .. 1e-6 of the updates)
                              gets that expression itself.  Decisions, draws and the cached log_post are the reference's at every one of those lane counts
                              (amwg_summation_order() == 1).  Every other geometry, and translated closures: the expression in every update, summed in lane order; a
                              closure with a row plan (amwg_user_model::rows_*) keeps the per-lane sums an update cannot have changed and prepares a sweep's
                              sums in one pass.
                              1 = every evaluation is the expression, with its full pass over the data, summed in the geometry's lane order (amwg_step_kernel).
                              2 = the hierarchical family's row layout without certified decisions (amwg_sweep_kernel): per-lane sums kept while an update cannot
                              have changed them (csrc/amwg_models.h lane_sum_rows), a sweep's proposed sums formed in one pass (prefetch_rows), its accept tests
                              one after the other from butterflies of those sums -- bit for bit what 1 computes.  A verification switch */
  int32_t test_bound_shift; /* TEST HOOK, 0 in production: the rounding bounds of the certified decisions (csrc/amwg_kernel.h: accept tests decided from a cheaper value
                               of log_post, from the local differences of a sweep) are multiplied by 2^shift, 0..40.  A wider bound sends more
                               updates down the path that evaluates the reference's expression; the results must not change by a bit (tests run 0 against 14 and 40) */
  int32_t sufficient_statistics; /* OPT-IN, 0 by default; the Normal family at one lane per chain only (AMWG_EINVAL elsewhere).  A THIRD TIER next to full_evaluation = 1
                               (the expression in every update, 8 operations per observation) and the default (the certified pass, 2 per observation): the cheaper
                               value of log_post the accept test is decided from (csrc/amwg_kernel.h "certified decisions") needs NO pass over the data for this
                               likelihood -- sum (x_i - mu)^2 = SS + n (xbar - mu)^2, with xbar (a double-double) and SS formed once on the host in quad precision --, so
                               an update costs the stepper alone, whatever n is.  Same bound, same fallback to the reference's expression, hence the same draws bit for
                               bit (tests/test_gpu_parity.py; tools/bound_audit.py audits it); what changes is that mcmc.js:524-526's pass over the observations is
                               made only where a uniform falls inside the bound.  Reported separately by bench.py, never as the headline */
} amwg_options;

typedef struct amwg_sampler amwg_sampler;

/* Replaces `new mcmc.AmwgSampler(params, log_post, data, options)` (mcmc.js:1090, 940-966).
 * init: P = sum(len) initial values (completed params' init, mcmc.js:954-957), the same for every chain. */
int amwg_create(const amwg_model_desc *model, const amwg_param_desc *params, int32_t n_params, const double *init,
                const amwg_comp_opt *comp_opts, const amwg_options *options, amwg_sampler **out);

/* A user-written `log_post(state, data)` closure (mcmc.js:958-960) translated to HIP by
 * bayes.js_amd/translate.js.  `source` defines `struct amwg::UserModel` (interface: csrc/amwg_kernel.h,
 * "translated closure"); amwg_create_user compiles it with hiprtc for the device's gfx target,
 * together with the same step kernel the built-in models use.  Arrays are the numeric arrays of
 * the closure's `data` argument that the body reads, flattened row-major. */
enum { AMWG_F64 = 0, AMWG_U8 = 1, AMWG_I32 = 2 };
typedef struct {
  const char *source;            /* HIP C++ text (NUL-terminated) */
  int32_t n_arrays;              /* any number (the first 16 pointers travel in the kernel arguments, the rest in a device table) */
  const double *const *arrays;   /* host pointers; copied to the device by amwg_create_user */
  const int64_t *array_len;      /* elements per array */
  const int32_t *array_type;     /* device storage per array: AMWG_F64 | AMWG_U8 | AMWG_I32 (values must be exactly representable);
                                    NULL = all AMWG_F64.  Host arrays are always doubles. */
  int32_t n_derived;             /* derived quantities (`state.key = expr`, mcmc.js:961-963, 990-995) recorded after the P components */
  int32_t lds_bytes;             /* bytes of data the generated stage() keeps in LDS (lanes_per_chain > 1) */
  int32_t lds_bytes_one_lane;    /* the same for lanes_per_chain == 1 (the generated code may stage other arrays then); 0 = same as lds_bytes */
  int32_t parallel;              /* 1 = the body has lane-split loops, lanes_per_chain > 1 is allowed */
  int32_t max_threads;           /* workgroup-size cap the translator suggests (0 = 1024) */
  double work_per_eval;          /* rough instruction count of one log_post evaluation (0 = unknown); only steers lanes_per_chain */
  double work_one_lane;          /* the same with ONE lane per chain, when the generated code then fast-forwards a two-valued sum (0 = no) */
  /* Row plan (csrc/amwg_rows.h): the closure ENDS in `lp += ld.norm(y[i], state.theta[g[i]], sd)` over all rows_n_obs observations, labels that repeat
   * with a stride of 64 and rows_groups <= 64 group means.  With 64 lanes per chain the library then lays the observations out in rows (one per lane)
   * and re-forms only the per-lane sums an update can have changed -- the same bits as evaluating everything (amwg_options::full_evaluation = 1
   * switches it off).  rows_sweep: the translator PROVED that a lane's sum depends on one entry of theta only, which allows the proposals of a whole
   * sweep over theta to be evaluated in one pass (UserModel::kRowSweep in the source says the same).  0 / 0 / 0 = no row plan.
   * The struct carries no size or version field: amwg_create_user therefore honours these three only as far as the GENERATED SOURCE states the same
   * (kRowN, kRowGroups, kRowSweep) and refuses a mismatch -- a caller built against an older, shorter struct cannot switch a layout on that the source has
   * no code for.  Everything the translator added since is read from the source alone: kRowCert (certified decisions in the row layout: amwg_user_sweep_cert),
   * kCertifiedTail / kTailN (certified decisions for a closure ending in a constant-mean normal loop: amwg_user_step_cert). */
  int32_t rows_n_obs, rows_groups, rows_sweep;
} amwg_user_model;

/* Replaces `new mcmc.AmwgSampler(params, log_post, data, options)` for an arbitrary (translated) closure. */
int amwg_create_user(const amwg_user_model *model, const amwg_param_desc *params, int32_t n_params, const double *init,
                     const amwg_comp_opt *comp_opts, const amwg_options *options, amwg_sampler **out);

/* hiprtc compilation of a translated closure without a device (build-time / CPU-test check).
 * arch e.g. "gfx950".  On failure returns AMWG_EINVAL and amwg_last_error() carries the compiler log. */
int amwg_compile_user(const char *source, int32_t lanes_per_chain, int32_t block_threads, const char *arch, size_t *code_bytes);

/* Compiled closures are kept on disk ($AMWG_CACHE_DIR, else $XDG_CACHE_HOME/amwg, else $HOME/.cache/amwg; AMWG_CACHE_DIR="" disables), keyed by
 * program text + kernel headers + compile options + target + hiprtc version: a second process constructing the same sampler loads the
 * code object instead of compiling it (README.md:41-42: a script that is simply run again).  This process's hits / misses so far; `dir`
 * (may be null) receives the directory in use, empty when the cache is off. */
int amwg_code_cache_stats(int64_t *hits, int64_t *misses, char *dir, size_t dir_capacity);

/* Replaces sampler.burn(n) (mcmc.js:1035-1039).  amwg_burn blocks until the steps are done;
 * amwg_burn_async only enqueues them on the sampler's stream (pair with amwg_sync), which is how
 * one host thread keeps several GPUs busy. */
int amwg_burn(amwg_sampler *s, int64_t n);
int amwg_burn_async(amwg_sampler *s, int64_t n);

/* Replaces sampler.sample(n) with thinning interval `thin` (mcmc.js:1005-1030, 1053-1055):
 * draw k is the state BEFORE step k*thin.  out_draws (host) receives ceil(n/thin) * P * chains
 * doubles laid out [draw][component][chain]; out_bytes is its capacity.  (P here and below means
 * amwg_num_recorded(): the parameters followed by a translated closure's derived quantities.) */
int amwg_sample(amwg_sampler *s, int64_t n, int64_t thin, double *out_draws, size_t out_bytes);

/* Two-phase form of amwg_sample for multi-GPU hosts: enqueue the steps into a library-owned
 * device buffer, then (after doing the same on the other devices) copy the draws out. */
int amwg_sample_async(amwg_sampler *s, int64_t n, int64_t thin);
int amwg_fetch_draws(amwg_sampler *s, double *out_draws, size_t out_bytes);
/* The same, delivered the way sampler.sample() returns it (mcmc.js:1009-1029: one array per monitored parameter): slice k receives the
 * recorded values base[k] .. base[k] + len[k] - 1 of every kept draw, laid out [draw][len[k]][chain], into out[k] (capacity out_bytes[k]).
 * Both forms copy the rows of each launch as soon as that launch has finished, while the later launches of the call still run.  The output
 * buffers must be writable and must not be touched by another thread during the call: their pages are made resident ahead of the copy by
 * rewriting one byte per page with its own value. */
int amwg_fetch_draws_slices(amwg_sampler *s, int32_t n_slices, const int32_t *base, const int32_t *len, double *const *out, const size_t *out_bytes);

/* Same, but the destination is DEVICE memory owned by the caller (e.g. a buffer that is then
 * gathered across GPUs with RCCL); no host copy is made.  Asynchronous on the sampler's stream
 * until amwg_sync(). */
int amwg_sample_device(amwg_sampler *s, int64_t n, int64_t thin, double *out_draws_dev, size_t out_bytes);

/* Replaces sampler.start_adaptation() / .stop_adaptation() (mcmc.js:1060-1073). */
int amwg_set_adapting(amwg_sampler *s, int32_t flag);

/* Replaces reading sampler.state (mcmc.js:964): out[component][chain], P*chains doubles. */
int amwg_get_state(amwg_sampler *s, double *out, size_t out_bytes);

/* Per-chain starting points (the reference starts every chain of a run at the same completed `init`,
 * mcmc.js:954-957; with many chains over-dispersed starts are what R-hat needs): state[component][chain],
 * P*chains doubles.  Invalidates the cached log_post, which is recomputed by the next launch. */
int amwg_set_state(amwg_sampler *s, const double *state, size_t state_bytes);

/* Replaces sampler.info() (mcmc.js:977-980 -> 906-912 -> 563-571).  Every array is
 * [component][chain]; any pointer may be NULL.  `accepts`/`inbounds` are run totals the
 * reference does not keep (accept decisions and in-bounds proposals), used by parity tests. */
int amwg_info(amwg_sampler *s, double *prop_log_scale, int32_t *acceptance_count, int32_t *iterations_since_adaption,
              int32_t *batch_count, int64_t *accepts, int64_t *inbounds);

/* Per chain: uniforms consumed so far, cached log_post(state), and the current order of the
 * named sub-steppers (mcmc.js:887), order[chain * n_params + k]. */
int amwg_chain_diag(amwg_sampler *s, uint64_t *uniforms, double *log_post, int32_t *named_order);

/* Posterior summaries computed on the device over the draws of the LAST amwg_sample* call:
 * mean[P], sd[P] (n-1 denominator) over all chains x kept draws. */
int amwg_last_sample_moments(amwg_sampler *s, double *mean, double *sd);

/* Convergence diagnostics over the draws of the LAST amwg_sample* call, computed on the device per recorded value
 * (what the reference's users do in R on the returned arrays, tests/test_mcmc_js.R): split-R-hat (every chain cut in
 * two halves; Gelman et al., BDA3 section 11.4) and the effective sample size from the variance of the chain means,
 * ess = chains * var_plus / Var(chain mean).  Needs >= 2 chains and >= 4 kept draws.  rhat[P], ess[P]. */
int amwg_last_sample_diagnostics(amwg_sampler *s, double *rhat, double *ess);

/* Posterior quantiles over the draws of the LAST amwg_sample* call (all chains x kept draws pooled), per recorded value:
 * radix sort on the device, R's default interpolation (type 7).  probs[n_probs] in [0,1]; out[P][n_probs]. */
int amwg_last_sample_quantiles(amwg_sampler *s, const double *probs, int32_t n_probs, double *out);

/* The same three summaries over SEVERAL samplers that are the shards of one logical job (chains split over the devices of a node
 * with amwg_options.chain_offset; `options.devices` of the JavaScript front-end): every device reduces its own draws and the
 * partial results are combined with an RCCL all-reduce over xGMI (quantiles: grouped ncclSend/ncclRecv of one component's
 * values to the first shard's device, sorted there) -- SURVEY.md section 8(e) "all-gather of per-chain moment summaries".
 * One process, ncclCommInitAll over the shards' devices (cached); RCCL is loaded on first use.  Shards may share a device
 * (they are summed on the device before the collective).  All shards must hold a sample() of the same number of kept draws.
 * Output sizes as for the single-sampler calls. */
int amwg_group_moments(amwg_sampler *const *shards, int32_t n_shards, double *mean, double *sd);
int amwg_group_diagnostics(amwg_sampler *const *shards, int32_t n_shards, double *rhat, double *ess);
int amwg_group_quantiles(amwg_sampler *const *shards, int32_t n_shards, const double *probs, int32_t n_probs, double *out);

/* The gather at sample collection (BASELINE.json north_star: "chains shard across the GPUs of one node with an RCCL-over-xGMI gather only at
 * sample collection").  The reference has one chain and nothing to gather: this is where `sampler.sample(n)`'s return
 * value (mcmc.js:1005-1030) is put together for a job whose chains live on several devices.
 *   amwg_group_gather_draws -- ONE process, one sampler per device: every shard's block of recorded draws [kept][P + derived][chains_i] of the
 *     last sample call travels to the device of shard `root_index` (grouped ncclSend / ncclRecv over the shards' communicator; shards on the
 *     root's own device are copied), where the blocks stand back to back in shard order -- in dst_device (on the root's device; may be null:
 *     a scratch buffer is used) and, if dst_host is given, in ONE copy to the host.  offsets[i] (optional) = first element of shard i's block.
 *   amwg_group_comm_info -- what the communicator says about itself: ranks (ncclCommCount) and the device of every rank. */
int amwg_group_gather_draws(amwg_sampler *const *shards, int32_t n_shards, int32_t root_index, double *dst_device, double *dst_host,
                            size_t capacity_bytes, int64_t *offsets);
int amwg_group_comm_info(amwg_sampler *const *shards, int32_t n_shards, int32_t *n_ranks, int32_t *devices, int32_t capacity);

/* The same exchanges for hosts that run ONE PROCESS PER DEVICE (torch.distributed.run, MPI, a process pool of Node workers): a communicator is
 * built from a shared 128-byte id -- amwg_comm_unique_id on one rank, the bytes travel by whatever the host has, amwg_comm_create on every
 * rank (collective) -- and then, all collective:
 *   amwg_comm_gather_draws -- the block [kept][P + derived][chains] of this rank's last sample call travels to rank `root`, where the blocks stand
 *     back to back in rank order in dst_device (root only).  Blocks may differ in size (uneven shards): the counts are exchanged first;
 *     counts (optional, one entry per rank, filled on every rank) = the elements each rank contributed.
 *   amwg_comm_moments -- mean and sd over the recorded draws of all ranks (two all-reduces of a few doubles), on every rank.
 *   amwg_comm_info -- ranks, own rank and device AS RCCL REPORTS THEM (ncclCommCount / ncclCommUserRank / ncclCommCuDevice). */
typedef struct amwg_comm amwg_comm;
#define AMWG_COMM_ID_BYTES 128
int amwg_comm_unique_id(char *id, size_t capacity);
int amwg_comm_create(const char *id, size_t id_bytes, int32_t n_ranks, int32_t rank, int32_t device, amwg_comm **out);
int amwg_comm_info(amwg_comm *c, int32_t *n_ranks, int32_t *rank, int32_t *device);
int amwg_comm_gather_draws(amwg_sampler *s, amwg_comm *c, int32_t root, double *dst_device, size_t capacity_bytes, int64_t *counts);
int amwg_comm_moments(amwg_sampler *s, amwg_comm *c, double *mean, double *sd);
int amwg_comm_destroy(amwg_comm *c);

/* AMWG_LANES_AUTOTUNE: the candidates that were timed at construction -- lanes[i] lanes per chain took ms[i] milliseconds PER STEP (the
 * fastest of several launches long enough to take >= 1 ms, after an untimed warm-up launch).  Returns the number of candidates (0 if the sampler was not autotuned); fills at most `cap` entries. */
int amwg_tuning(const amwg_sampler *s, int32_t *lanes, double *ms, int32_t cap);

int amwg_sync(amwg_sampler *s);
int amwg_num_components(const amwg_sampler *s);   /* P: scalar parameter components */
int amwg_num_recorded(const amwg_sampler *s);     /* values per draw row: P + derived quantities */
int64_t amwg_num_chains(const amwg_sampler *s);
/* Launch geometry actually used and HIP-event time of the step kernels of the last burn/sample call. */
int amwg_launch_info(const amwg_sampler *s, int32_t *lanes_per_chain, int32_t *block_threads, int32_t *grid_blocks,
                     int32_t *lds_bytes, int32_t *n_launches, double *kernel_ms);
/* Name of the step kernel this sampler launches, as a profiler lists it (without the amwg:: qualifiers): "amwg_step_kernel<HierNormalModel,64,512>",
 * "amwg_sweep_kernel<HierNormalModel,512>" (the hierarchical family's row layout: lane-local re-evaluation + sweep prefetch), "amwg_step_kernel_cert<NormalModel,1,256>" /
 * "amwg_sweep_kernel_cert<HierNormalModel,512>" (the kernels that decide from certified values: options.full_evaluation = 0 where a family has them),
 * "amwg_gl_kernel<HierGlModel,512>" (options.group_local), "amwg_user_step" (a translated closure).  The last number is the workgroup size
 * class the kernel was compiled for (256 / 512 / 1024).  Valid until the sampler is destroyed. */
const char *amwg_kernel_name(const amwg_sampler *s);
/* The summation order this sampler's decisions and its cached log_post follow: the number of per-lane partial sums log_post is formed from.
 * 1 = the REFERENCE's own order (one running sum over the closure's terms, mcmc.js:524-526 calling the model's log_post): accept counts and draws are
 * the reference's bit for bit -- every sampler at one lane per chain, and at any lane count the kernels that decide from certified values against the
 * expression in that order (round 5: the hierarchical family's sweep kernel at 64 lanes, the Poisson family at 16 lanes; options.full_evaluation = 0).
 * Otherwise lanes_per_chain: the expression is summed per lane and the lanes' sums by a butterfly (the oracle's lane order; a decision can differ from
 * the reference's where a uniform falls between the two orders' values, ~1e-10 of the decisions).  Negative: an error code. */
int amwg_summation_order(const amwg_sampler *s);
int amwg_destroy(amwg_sampler *s);
const char *amwg_last_error(void);
const char *amwg_version(void);

/* Host evaluations of the exact source the kernel compiles (csrc/amwg_math.h, amwg_philox.h): what the JavaScript front-end's host-side
 * log_post() and the data generators of the harnesses use.  (The remaining building blocks are exported one by one by the test build
 * only: include/amwg_selftest.h, libamwg_selftest.so.) */
double amwg_exp(double x);   /* bit-identical to V8 Math.exp */
double amwg_log(double x);   /* bit-identical to V8 Math.log */
double amwg_uniform(uint64_t seed, uint64_t chain, uint64_t index);
/* Measured fp64 vector issue ceiling of the device: a register-only kernel of independent v_fma_f64 chains on every SIMD
 * (what the chip sustains under fp64 load at its own clocks), in lane-operations per second.  bench.py prices the step
 * kernel against this next to the datasheet number. */
int amwg_fp64_peak(int32_t device, double *lane_ops_per_s);

#ifdef __cplusplus
}
#endif
#endif
