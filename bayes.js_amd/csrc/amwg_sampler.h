// amwg_sampler.h -- the sampler handle behind the C ABI (internal to libamwg.so; shared by amwg_core.hip and
// amwg_summaries.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../include/amwg.h"
#include "amwg_types.h"

typedef void (*step_kernel_t)(const amwg::StepArgs);

struct amwg_sampler {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int model = 0, P = 0, n_params = 0;
  int64_t C = 0;
  amwg_options opt{};
  amwg::ParamLayout pl{};
  amwg::ModelConsts mc{};
  amwg::DataRef d{};
  amwg::ChainArrays ch{};
  std::vector<void *> dev_allocs;
  amwg::CompConst *d_cc = nullptr;
  uint8_t *d_adapt = nullptr;
  std::vector<uint8_t> h_adapt;
  std::vector<int32_t> h_layout;   // [4][n_params] base | len | top | multidim (ParamLayout::tab on the device)
  // geometry
  int lanes = 0, block = 0, grid = 0, lds = 0, cpb = 0;   // cpb: chains per workgroup if fewer than block / lanes (StepArgs::cpb)
  step_kernel_t kernel = nullptr;
  bool certified = false;      // `kernel` is one of the kernels that decide from certified values (amwg_kernel.h kCert: amwg_step_kernel_cert / amwg_sweep_kernel_cert)
  std::string kernel_name;          // amwg_kernel_name(): filled on first request
  uint32_t hier_periodic_mask = 0;   // HIER: bit j set = the group labels repeat with a lane stride of 2^j (g[i] == g[i mod 2^j])
  bool lp_ready = false;
  bool lp_is_expression = true;      // ch.lp_curr holds the expression's value for every chain (false after steps of a kernel with certified decisions, until a finalize launch)
  std::vector<std::pair<int, float>> tuned;   // AMWG_LANES_AUTOTUNE: (lanes per chain, ms of the timing run) of every candidate
  // translated closure (amwg_create_user): hiprtc module function instead of a built-in kernel
  bool user = false;
  int D = 0;                       // derived quantities recorded after the P components
  int user_lds = 0, user_lds_one_lane = 0, user_parallel = 0, user_max_threads = 1024;
  int user_rows_n = 0, user_rows_groups = 0, user_rows_sweep = 0;   // row plan of a translated closure (amwg_rows.h): observations and groups of its final likelihood loop; 0 = none
  bool user_rows_cert = false;     // the row plan has certified values (kRowCert of the generated source: amwg_rows.h log_post_approx / sweep_approx / reference_order)
  int user_cert_tail_n = 0;        // certified tail of a translated closure (amwg_user.h norm_tail_approx): observations of its final constant-mean normal loop (kTailN of the generated source); 0 = none
  int user_pois_tail_n = 0;        // certified Poisson tail of a translated closure (amwg_ptail.h pois_tail_approx): observations of its final log-link Poisson loop (kPoisTail / kTailN of the generated source); 0 = none
  bool user_sweep = false, user_has_binary = false;                  // the chosen geometry runs amwg_user_sweep; the model has binary parameters
  double user_work = 0;            // translator's estimate of the instructions of one log_post evaluation
  double user_work_one_lane = 0;   // the same with one lane per chain when that enables a fast-forwarded sum (0 = n/a)
  hipFunction_t user_fn = nullptr;
  hipModule_t user_module = nullptr;
  // last call
  int n_launches = 0;
  double kernel_ms = 0.0;
  double *d_draws = nullptr;       // library-owned draw buffer of the last amwg_sample
  size_t d_draws_cap = 0;
  const double *last_draws = nullptr;  // device pointer (library- or caller-owned) of the last sample call
  int64_t last_rows = 0;
  // rows of the last sample call become final launch by launch: an event after each launch and the row count up to it, so that
  // amwg_fetch_draws* copies the rows of launch j on copy_stream while launches j + 1, ... still run
  std::vector<hipEvent_t> chunk_ev;
  std::vector<int64_t> chunk_rows;     // cumulative rows after launch j of the last sample call
  hipStream_t copy_stream = nullptr;
};

// shared helper: records an error message for amwg_last_error() and returns `code`
int amwg_fail(int code, const char *fmt, ...);
