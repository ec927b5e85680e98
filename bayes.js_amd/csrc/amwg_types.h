// amwg_types.h -- plain structs shared by the host core and the kernels.
#pragma once
#include "amwg_stdint.h"

namespace amwg {

// No fixed limits on the model's shape (the reference has none: mcmc.js:837-881 takes any number of named parameters, mcmc.js:631-680
// any dim); only the layout's fast paths have sizes:
constexpr int kPackedNamed = 16;     // up to 16 named parameters: their shuffled order is sixteen 4-bit fields of one u64 in a register;
                                     // more: 16-bit entries in LDS (ChainArrays::perm16)
constexpr int kByteTop = 256;        // a shuffled leading dimension up to 256: index bytes in LDS; longer: 16-bit entries
constexpr int kMaxIndex = 65535;     // 16-bit entries: at most 65535 named parameters / leading dimension (LDS capacity binds long before)
constexpr int kInlineUserArrays = 16;   // data arrays of a translated (user) log_post passed in the kernel arguments; further ones
                                        // through DataRef::arr_ext (a pointer table in device memory)
constexpr int kTypeReal = 0, kTypeInt = 1, kTypeBinary = 2, kTypeFixed = 3;   // AMWG_REAL / AMWG_INT / AMWG_BINARY / AMWG_FIXED

// LDS-resident view of this chain's state: component p at S.base[p].  Chains are laid out
// [chain][stride] with an ODD stride (in doubles): lanes that own different chains and read the
// same component hit 32 distinct 8-byte bank slots, and the G lanes of one chain that gather
// different components (theta[g_i]) read consecutive addresses -- conflict-free both ways.
struct StateView {
  const double *base;
#if defined(__HIPCC__)
  __host__ __device__ __forceinline__
#else
  inline
#endif
  double operator()(int p) const { return base[p]; }
};

// Per scalar component, identical for all chains (mcmc.js:497-505).
struct CompConst {
  double lower, upper, max_adaptation, initial_adaptation, target_accept_rate;
  double batch_size;   // a JavaScript number (mcmc.js:538, 543 compare and divide by it as such)
  int32_t type;  // AMWG_REAL / AMWG_INT / AMWG_BINARY
};

// Completed parameter layout (mcmc.js:357-403), flattened.
// The first n_params named parameters (P_stepped scalar components) are stepped; state entries P_stepped..P-1 are only read
// by log_post (AMWG_FIXED: the rest of the shared state object of a stand-alone stepper).
struct ParamLayout {
  int32_t n_params, P, max_top, P_stepped;
  const int32_t *tab;     // device: [4][n_params] = base | len | top | multidim of every stepped parameter (staged into LDS per launch)
};

// Loop-invariant constants of the built-in models, computed ON THE HOST with the same
// log_v8 the kernel uses (so the roundings are those of the reference's expression trees).
struct ModelConsts {
  double neg_half_log_2pi;          // -0.5*log(2*pi)
  // prior of the location parameter:  ld.norm(v, m0, s0) = c0 - (v-m0)^2 / den0
  double m0, c0, den0;
  // prior of the scale parameter:     ld.unif(v, ua, ub)
  double ua, ub, lunif;
  // second-level normal prior with constant sd tau (HIER: theta_g ~ norm(mu, tau)): c1 - (v-mu)^2 / den1
  double c1, den1;
  // ld.beta(theta, ba, bb)
  double ba, bb, lbeta_ab;
  // ld.unif(cp, 0, N-1)
  double cp_upper, lunif_cp;
  // double-double reciprocals of the constant prior divisors den0 / den1 (amwg_div.h; {hi, lo}) and whether they are usable
  double y0_hi, y0_lo, y1_hi, y1_lo;
  int32_t den0_ok, den1_ok;
  int32_t data_mid_range;           // every data value is 0 or within 2^-200..2^200 in magnitude
  int32_t exact_division;           // 1 = always use IEEE '/'
  int32_t has_invalid;              // BETA_BERN: some x_i is neither 0 nor 1 => that term is -inf (distributions.js:229)
  int32_t group_local;              // HIER: group-local evaluation (amwg_options::group_local; preconditions checked by amwg_create)
  // POIS_GLM, for the bounds of the certified pass (amwg_models.h PoisGlmModel::log_post_approx): max_i |X[i][k]| per column, sum of the counts, sum of
  // lfactorial(y_i) (+inf if some count is negative: such data always takes the expression)
  double glm_xmax[7], glm_sum_y, glm_sum_lf;
  // NORMAL, amwg_options::sufficient_statistics: sum (x_i - mu)^2 = suff_ss + n (xbar - mu)^2, xbar = suff_xbar_hi + suff_xbar_lo (formed on the host in quad precision)
  double suff_xbar_hi, suff_xbar_lo, suff_ss;
  int32_t sufficient;
  int32_t group_lane_const;         // HIER: g[i] == g[i % lanes] for every i -- each lane of a chain only ever meets ONE group (balanced
                                    // round-robin designs such as g_i = i mod 32 with 64 lanes): its mean is read once per evaluation
};

// Device pointers to the (read-only, chain-shared) data.
struct DataRef {
  int32_t n_obs, G, K, pad;
  const double *x;      // NORMAL x[N] | HIER y[N] | GLM X column-major [K][N] (coalesced across observations)
  const double *y;      // GLM counts
  const double *lfact;  // GLM lfactorial(y_i) (+inf encodes y_i < 0, i.e. term = -inf)
  const uint8_t *xb;    // BETA_BERN x as bytes (invalid values stored as 0, see has_invalid) | HIER group index
  const uint32_t *xw;   // BETA_BERN x as bits: observation i = bit (i & 31) of word (i >> 5)
  const void *arr[kInlineUserArrays];  // translated models: the data arrays the closure reads (row-major; f64, or u8 / i32
                                       // when every value of the array is a small integer -- the translator picks the type)
  const void *const *arr_ext;          // arrays kInlineUserArrays, kInlineUserArrays + 1, ... (device table), see user_arr()
  double *wave_scratch;                // 64 doubles per wavefront of the launch (device memory): where the certified pass of the Normal family at one lane per chain
                                       // leaves the wavefront's 64 means for the scalar memory path (amwg_pass.h norm_sq_pass_wave); nullptr = v_readlane
};

// Per-chain state, structure-of-arrays with the chain index fastest: element (p, c) at p*C + c.
struct ChainArrays {
  double *state;            // [P][C] current value of every scalar component
  double *prop_log_scale;   // [P][C]
  int32_t *acceptance_count, *iterations_since_adaption, *batch_count;  // [P][C]  (mcmc.js:509-511)
  int32_t *accepts, *inbounds;   // [P][C] run totals (not in the reference; for parity checks)
  uint64_t *perm;           // [C] order of the named sub-steppers, 4 bits each (mcmc.js:887 shuffles in place); n_params <= kPackedNamed
  uint16_t *perm16;         // [n_params][C] the same order as 16-bit entries when n_params > kPackedNamed (else null)
  uint64_t *rng_n;          // [C] uniforms consumed
  double *lp_curr;          // [C] log_post(state): the expression's value -- or, between the launches of a kernel with certified decisions, the cheaper value the
                            // stepper last decided from, when lp_eps says so (made exact on demand: StepArgs::finalize_lp)
  double *lp_eps;           // [C] 0: lp_curr is the expression's value; else the bound that goes with the cheaper value in lp_curr
  double *audit;            // BOUND AUDIT builds only (libamwg_audit.so, -DAMWG_AUDIT; null in the product): [4][C] per chain -- max |A - E| / eps over the audited values,
                            // max |dA - dE| / eta over the audited differences, audited decisions, decisions a certified verdict got WRONG (amwg_kernel.h "BOUND AUDIT")
  unsigned long long *audit_hist;   // ... and [2][64] counts of those ratios by binary exponent (bin 40 = [1, 2): a violated bound)
  int32_t *error;           // one word: bits set by a step kernel that refused its launch or found its own bookkeeping inconsistent (amwg_kernel.h
                            // device_error); the host reads it after every call and turns it into an error -- never a silent no-op
};

struct StepArgs {
  int64_t C;                // chains on this device
  uint64_t seed, chain_offset;
  int32_t n_steps, thin;
  int64_t step0;            // index of the first step of this launch inside the sample() call (for i % thin)
  double *draws;            // nullptr for burn(); else [row][P][C]
  int64_t row0;             // first draw row this launch writes
  const CompConst *cc;      // [P]
  const uint8_t *is_adapting;  // [P]
  int32_t init_lp;          // 1 = first launch: compute lp_curr = log_post(init) (the ctor's warm-up call, mcmc.js:961-963)
  int32_t finalize_lp;              // 1 = leave the expression's value of log_post(state) in lp_curr (amwg_chain_diag asks for it; a launch otherwise hands the stepper's
                                    // cheaper value and its bound on to the next launch: no closing evaluation per launch)
  int32_t certified;                // 1 = this launch is of a kernel that decides accept tests from a model's cheaper value of log_post with its bound (amwg_kernel.h
                                    // "certified decisions": amwg_step_kernel_cert / amwg_sweep_kernel_cert).  Informational: the kernels do not branch on it
  double bound_scale;               // 2^amwg_options::test_bound_shift (1 in production): multiplies the bounds of the certified decisions
  int32_t audit_adversarial;        // BOUND AUDIT build only (0 in the product, which does not read it): the accept uniform of every certified decision is REPLACED by one placed
                                    // 1.5 eta off exp(dA) -- just outside the sliver, alternately on either side --, the worst case for a bound that is too small (tools/bound_audit.py --shrink)
  int32_t sweep_update_by_update;   // amwg_options::full_evaluation == 2: the sweep kernel decides a sweep's accept tests one after the other (verification switch)
  int32_t cpb;              // chains per workgroup when the per-chain state of blockDim / lanes chains does not fit LDS (0 = all of them);
                            // the lane groups beyond cpb then replicate the workgroup's last chain (same stream, same stores)
  ParamLayout pl;
  ModelConsts mc;
  DataRef d;
  ChainArrays ch;
};

}  // namespace amwg
