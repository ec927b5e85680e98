/*
 * amwg_napi.c -- Node N-API shim over the C ABI of include/amwg.h.
 *
 * This is the binding a bayes.js maintainer adds: the reference is pure JS
 * with no FFI, so the JS front-end (bayes.js_amd/mcmc.js) keeps the AmwgSampler API
 * (mcmc.js:1090-1099) and forwards to these functions.  Marshalling only: typed arrays in,
 * typed arrays out, errors from amwg_last_error() rethrown as JS Errors.  All calls are
 * synchronous on the single JS thread, like the reference.
 */
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/amwg.h"

#define NAPI_OK(call)                                                   \
  do {                                                                  \
    if ((call) != napi_ok) {                                            \
      napi_throw_error(env, NULL, "amwg_napi: N-API call failed: " #call); \
      return NULL;                                                      \
    }                                                                   \
  } while (0)

static napi_value throw_amwg(napi_env env, int rc) {
  char buf[600];
  snprintf(buf, sizeof buf, "amwg error %d: %s", rc, amwg_last_error());
  napi_throw_error(env, rc == AMWG_EHIP ? "AMWG_EHIP" : (rc == AMWG_ESIZE ? "AMWG_ESIZE" : "AMWG_EINVAL"), buf);
  return NULL;
}

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv) {
  size_t argc = want;
  if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
    napi_throw_type_error(env, NULL, "amwg_napi: wrong number of arguments");
    return 0;
  }
  return 1;
}

static int prop(napi_env env, napi_value obj, const char *name, napi_value *out) {
  bool has = false;
  if (napi_has_named_property(env, obj, name, &has) != napi_ok || !has) return 0;
  if (napi_get_named_property(env, obj, name, out) != napi_ok) return 0;
  napi_valuetype t;
  if (napi_typeof(env, *out, &t) != napi_ok || t == napi_undefined || t == napi_null) return 0;
  return 1;
}

static double prop_double(napi_env env, napi_value obj, const char *name, double dflt) {
  napi_value v;
  double d = dflt;
  if (prop(env, obj, name, &v)) napi_get_value_double(env, v, &d);
  return d;
}

static int64_t prop_i64(napi_env env, napi_value obj, const char *name, int64_t dflt) {
  napi_value v;
  if (!prop(env, obj, name, &v)) return dflt;
  double d = (double)dflt;
  napi_valuetype t;
  napi_typeof(env, v, &t);
  if (t == napi_boolean) { bool b = false; napi_get_value_bool(env, v, &b); return b ? 1 : 0; }
  napi_get_value_double(env, v, &d);
  return (int64_t)d;
}

/* unsigned 64-bit from a JS number (exact below 2^53) or a BigInt */
static uint64_t prop_u64(napi_env env, napi_value obj, const char *name, uint64_t dflt) {
  napi_value v;
  if (!prop(env, obj, name, &v)) return dflt;
  napi_valuetype t;
  napi_typeof(env, v, &t);
  if (t == napi_bigint) {
    uint64_t u = dflt;
    bool lossless = true;
    napi_get_value_bigint_uint64(env, v, &u, &lossless);
    return u;
  }
  double d = 0;
  napi_get_value_double(env, v, &d);
  return d <= 0 ? 0 : (uint64_t)d;
}

static void *typed_data(napi_env env, napi_value v, napi_typedarray_type want, size_t *len) {
  bool is = false;
  if (napi_is_typedarray(env, v, &is) != napi_ok || !is) return NULL;
  napi_typedarray_type t;
  void *data = NULL;
  napi_value ab;
  size_t off;
  if (napi_get_typedarray_info(env, v, &t, len, &data, &ab, &off) != napi_ok || t != want) return NULL;
  return data;
}

static napi_value new_f64(napi_env env, size_t n, double **data) {
  napi_value ab, ta;
  void *p = NULL;
  if (napi_create_arraybuffer(env, n * 8, &p, &ab) != napi_ok) return NULL;
  if (napi_create_typedarray(env, napi_float64_array, n, ab, 0, &ta) != napi_ok) return NULL;
  *data = (double *)p;
  return ta;
}

static napi_value new_i32(napi_env env, size_t n, int32_t **data) {
  napi_value ab, ta;
  void *p = NULL;
  if (napi_create_arraybuffer(env, n * 4, &p, &ab) != napi_ok) return NULL;
  if (napi_create_typedarray(env, napi_int32_array, n, ab, 0, &ta) != napi_ok) return NULL;
  *data = (int32_t *)p;
  return ta;
}

static void finalize_sampler(napi_env env, void *data, void *hint) {
  (void)env; (void)hint;
  amwg_sampler **box = (amwg_sampler **)data;
  if (box) { if (*box) amwg_destroy(*box); free(box); }
}

static amwg_sampler *unwrap(napi_env env, napi_value v) {
  amwg_sampler **box = NULL;
  if (napi_get_value_external(env, v, (void **)&box) != napi_ok || !box || !*box) {
    napi_throw_error(env, NULL, "amwg_napi: sampler handle is closed or invalid");
    return NULL;
  }
  return *box;
}

static int64_t arg_i64(napi_env env, napi_value v);

/* params[], init, compOpts[], options -> C descriptors (caller frees *pd and *co) */
static int parse_common(napi_env env, napi_value *a /* [1]=params [2]=init [3]=compOpts [4]=options */, amwg_param_desc **pd_out,
                        amwg_comp_opt **co_out, uint32_t *n_params_out, const double **init_out, amwg_options *op) {
  uint32_t n_params = 0, n_comp = 0;
  if (napi_get_array_length(env, a[1], &n_params) != napi_ok || napi_get_array_length(env, a[3], &n_comp) != napi_ok) {
    napi_throw_type_error(env, NULL, "amwg_napi.create: params and compOpts must be arrays");
    return 0;
  }
  size_t n_init = 0;
  const double *init = (const double *)typed_data(env, a[2], napi_float64_array, &n_init);
  if (!init || n_init != n_comp || n_params < 1) {
    napi_throw_type_error(env, NULL, "amwg_napi.create: init must be a Float64Array with one value per component");
    return 0;
  }
  amwg_param_desc *pd = (amwg_param_desc *)calloc(n_params, sizeof *pd);
  amwg_comp_opt *co = (amwg_comp_opt *)calloc(n_comp, sizeof *co);
  for (uint32_t i = 0; i < n_params; i++) {
    napi_value e;
    napi_get_element(env, a[1], i, &e);
    pd[i].type = (int32_t)prop_i64(env, e, "type", 0);
    pd[i].len = (int32_t)prop_i64(env, e, "len", 1);
    pd[i].top = (int32_t)prop_i64(env, e, "top", 1);
    pd[i].multidim = (int32_t)prop_i64(env, e, "multidim", 0);
    pd[i].lower = prop_double(env, e, "lower", -1.0 / 0.0);
    pd[i].upper = prop_double(env, e, "upper", 1.0 / 0.0);
  }
  for (uint32_t i = 0; i < n_comp; i++) {
    napi_value e;
    napi_get_element(env, a[3], i, &e);
    co[i].prop_log_scale = prop_double(env, e, "prop_log_scale", 0.0);
    co[i].max_adaptation = prop_double(env, e, "max_adaptation", 0.33);
    co[i].initial_adaptation = prop_double(env, e, "initial_adaptation", 1.0);
    co[i].target_accept_rate = prop_double(env, e, "target_accept_rate", 0.44);
    co[i].batch_size = prop_double(env, e, "batch_size", 50.0);
    co[i].is_adapting = (int32_t)prop_i64(env, e, "is_adapting", 1);
  }
  memset(op, 0, sizeof *op);
  op->chains = prop_i64(env, a[4], "chains", 1);
  op->seed = prop_u64(env, a[4], "seed", 0);
  op->chain_offset = prop_u64(env, a[4], "chain_offset", 0);
  op->device = (int32_t)prop_i64(env, a[4], "device", 0);
  op->lanes_per_chain = (int32_t)prop_i64(env, a[4], "lanes_per_chain", 0);
  op->block_threads = (int32_t)prop_i64(env, a[4], "block_threads", 0);
  op->steps_per_launch = (int32_t)prop_i64(env, a[4], "steps_per_launch", 0);
  op->exact_division = (int32_t)prop_i64(env, a[4], "exact_division", 0);
  op->group_local = (int32_t)prop_i64(env, a[4], "group_local", 0);
  op->full_evaluation = (int32_t)prop_i64(env, a[4], "full_evaluation", 0);
  op->test_bound_shift = (int32_t)prop_i64(env, a[4], "test_bound_shift", 0);
  op->sufficient_statistics = (int32_t)prop_i64(env, a[4], "sufficient_statistics", 0);
  *pd_out = pd; *co_out = co; *n_params_out = n_params; *init_out = init;
  return 1;
}

static napi_value wrap_sampler(napi_env env, amwg_sampler *s) {
  amwg_sampler **box = (amwg_sampler **)malloc(sizeof *box);
  *box = s;
  napi_value ext;
  if (napi_create_external(env, box, finalize_sampler, NULL, &ext) != napi_ok) {
    amwg_destroy(s);
    free(box);
    napi_throw_error(env, NULL, "amwg_napi: napi_create_external failed");
    return NULL;
  }
  return ext;
}

/* create(model, params[], init Float64Array, compOpts[], options) -> external handle */
static napi_value Create(napi_env env, napi_callback_info info) {
  napi_value a[5];
  if (!get_args(env, info, 5, a)) return NULL;
  amwg_model_desc md;
  memset(&md, 0, sizeof md);
  md.model = (int32_t)prop_i64(env, a[0], "model", 0);
  md.n_obs = (int32_t)prop_i64(env, a[0], "n_obs", 0);
  md.G = (int32_t)prop_i64(env, a[0], "G", 0);
  md.K = (int32_t)prop_i64(env, a[0], "K", 0);
  napi_value v;
  size_t n = 0;
  if (prop(env, a[0], "x", &v)) md.x = (const double *)typed_data(env, v, napi_float64_array, &n);
  if (prop(env, a[0], "y", &v)) md.y = (const double *)typed_data(env, v, napi_float64_array, &n);
  if (prop(env, a[0], "g", &v)) md.g = (const int32_t *)typed_data(env, v, napi_int32_array, &n);
  if (prop(env, a[0], "hyper", &v)) {
    const double *h = (const double *)typed_data(env, v, napi_float64_array, &n);
    for (size_t i = 0; h && i < n && i < 8; i++) md.hyper[i] = h[i];
  }
  amwg_param_desc *pd; amwg_comp_opt *co; uint32_t n_params; const double *init; amwg_options op;
  if (!parse_common(env, a, &pd, &co, &n_params, &init, &op)) return NULL;
  amwg_sampler *s = NULL;
  int rc = amwg_create(&md, pd, (int32_t)n_params, init, co, &op, &s);
  free(pd);
  free(co);
  if (rc != AMWG_OK) return throw_amwg(env, rc);
  return wrap_sampler(env, s);
}

/* createUser({source, arrays: [Float64Array...], n_derived, lds_bytes, parallel, max_threads}, params[], init, compOpts[], options)
 * -- a closure translated by bayes.js_amd/translate.js (amwg_create_user) */
static napi_value CreateUser(napi_env env, napi_callback_info info) {
  napi_value a[5];
  if (!get_args(env, info, 5, a)) return NULL;
  napi_value v;
  if (!prop(env, a[0], "source", &v)) { napi_throw_type_error(env, NULL, "amwg_napi.createUser: source missing"); return NULL; }
  size_t slen = 0;
  if (napi_get_value_string_utf8(env, v, NULL, 0, &slen) != napi_ok) { napi_throw_type_error(env, NULL, "amwg_napi.createUser: source must be a string"); return NULL; }
  char *src = (char *)malloc(slen + 1);
  napi_get_value_string_utf8(env, v, src, slen + 1, &slen);
  uint32_t n_arr = 0;
  if (prop(env, a[0], "arrays", &v)) napi_get_array_length(env, v, &n_arr);
  /* any number of data arrays (a closure over an array of records reads one per column) */
  const double **arrs = (const double **)calloc(n_arr ? n_arr : 1, sizeof *arrs);
  int64_t *lens = (int64_t *)calloc(n_arr ? n_arr : 1, sizeof *lens);
  int32_t *types = (int32_t *)calloc(n_arr ? n_arr : 1, sizeof *types);
  if (!arrs || !lens || !types) { free(src); free(arrs); free(lens); free(types); napi_throw_error(env, NULL, "amwg_napi.createUser: out of memory"); return NULL; }
  for (uint32_t i = 0; i < n_arr; i++) {
    napi_value e;
    size_t n = 0;
    napi_get_element(env, v, i, &e);
    arrs[i] = (const double *)typed_data(env, e, napi_float64_array, &n);
    lens[i] = (int64_t)n;
    if (!arrs[i] && n) { free(src); free(arrs); free(lens); free(types); napi_throw_type_error(env, NULL, "amwg_napi.createUser: arrays must be Float64Arrays"); return NULL; }
  }
  amwg_user_model um;
  memset(&um, 0, sizeof um);
  um.source = src;
  um.n_arrays = (int32_t)n_arr;
  um.arrays = arrs;
  um.array_len = lens;
  um.array_type = types;
  {
    napi_value tv;
    uint32_t nt = 0;
    if (prop(env, a[0], "array_types", &tv) && napi_get_array_length(env, tv, &nt) == napi_ok)
      for (uint32_t i = 0; i < nt && i < n_arr; i++) { napi_value e; napi_get_element(env, tv, i, &e); types[i] = (int32_t)arg_i64(env, e); }
  }
  um.n_derived = (int32_t)prop_i64(env, a[0], "n_derived", 0);
  um.lds_bytes = (int32_t)prop_i64(env, a[0], "lds_bytes", 0);
  um.lds_bytes_one_lane = (int32_t)prop_i64(env, a[0], "lds_bytes_one_lane", 0);
  um.parallel = (int32_t)prop_i64(env, a[0], "parallel", 0);
  um.max_threads = (int32_t)prop_i64(env, a[0], "max_threads", 0);
  um.work_per_eval = prop_double(env, a[0], "work_per_eval", 0.0);
  um.work_one_lane = prop_double(env, a[0], "work_one_lane", 0.0);
  um.rows_n_obs = (int32_t)prop_double(env, a[0], "rows_n_obs", 0.0);      /* row plan (csrc/amwg_rows.h), 0 = none */
  um.rows_groups = (int32_t)prop_double(env, a[0], "rows_groups", 0.0);
  um.rows_sweep = (int32_t)prop_double(env, a[0], "rows_sweep", 0.0);
  amwg_param_desc *pd; amwg_comp_opt *co; uint32_t n_params; const double *init; amwg_options op;
  if (!parse_common(env, a, &pd, &co, &n_params, &init, &op)) { free(src); free(arrs); free(lens); free(types); return NULL; }
  amwg_sampler *s = NULL;
  int rc = amwg_create_user(&um, pd, (int32_t)n_params, init, co, &op, &s);
  free(pd);
  free(co);
  free(src);
  free(arrs);
  free(lens);
  free(types);
  if (rc != AMWG_OK) return throw_amwg(env, rc);
  return wrap_sampler(env, s);
}

/* compileUser(source, lanes, block, arch) -> code object size; throws with the hiprtc log (no device needed) */
static napi_value CompileUser(napi_env env, napi_callback_info info) {
  napi_value a[4], r;
  if (!get_args(env, info, 4, a)) return NULL;
  size_t slen = 0, alen = 0;
  napi_get_value_string_utf8(env, a[0], NULL, 0, &slen);
  char *src = (char *)malloc(slen + 1);
  napi_get_value_string_utf8(env, a[0], src, slen + 1, &slen);
  char arch[64];
  napi_get_value_string_utf8(env, a[3], arch, sizeof arch, &alen);
  size_t bytes = 0;
  int rc = amwg_compile_user(src, (int32_t)arg_i64(env, a[1]), (int32_t)arg_i64(env, a[2]), arch, &bytes);
  free(src);
  if (rc != AMWG_OK) return throw_amwg(env, rc);
  NAPI_OK(napi_create_double(env, (double)bytes, &r));
  return r;
}

static napi_value Destroy(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  amwg_sampler **box = NULL;
  if (napi_get_value_external(env, a[0], (void **)&box) == napi_ok && box && *box) {
    amwg_destroy(*box);
    *box = NULL;
  }
  return NULL;
}

static int64_t arg_i64(napi_env env, napi_value v) {
  double d = 0;
  napi_get_value_double(env, v, &d);
  return (int64_t)d;
}

static napi_value Burn(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  amwg_sampler *s = unwrap(env, a[0]);
  if (!s) return NULL;
  int rc = amwg_burn(s, arg_i64(env, a[1]));
  return rc == AMWG_OK ? NULL : throw_amwg(env, rc);
}

static napi_value BurnAsync(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  amwg_sampler *s = unwrap(env, a[0]);
  if (!s) return NULL;
  int rc = amwg_burn_async(s, arg_i64(env, a[1]));
  return rc == AMWG_OK ? NULL : throw_amwg(env, rc);
}

static napi_value Sync(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  amwg_sampler *s = unwrap(env, a[0]);
  if (!s) return NULL;
  int rc = amwg_sync(s);
  return rc == AMWG_OK ? NULL : throw_amwg(env, rc);
}

static napi_value SampleAsync(napi_env env, napi_callback_info info) {
  napi_value a[3];
  if (!get_args(env, info, 3, a)) return NULL;
  amwg_sampler *s = unwrap(env, a[0]);
  if (!s) return NULL;
  int rc = amwg_sample_async(s, arg_i64(env, a[1]), arg_i64(env, a[2]));
  return rc == AMWG_OK ? NULL : throw_amwg(env, rc);
}

/* fetchDraws(handle, rows) -> Float64Array [rows][P][chains] */
static napi_value FetchDraws(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  amwg_sampler *s = unwrap(env, a[0]);
  if (!s) return NULL;
  const int64_t rows = arg_i64(env, a[1]);
  const size_t n = (size_t)rows * (size_t)amwg_num_recorded(s) * (size_t)amwg_num_chains(s);
  double *data = NULL;
  napi_value out = new_f64(env, n, &data);
  if (!out) { napi_throw_error(env, NULL, "amwg_napi: cannot allocate the draws array"); return NULL; }
  int rc = amwg_fetch_draws(s, data, n * 8);
  return rc == AMWG_OK ? out : throw_amwg(env, rc);
}

/* fetchDrawsSplit(handle, rows, bases, lens) -> [Float64Array [rows][lens[k]][chains], ...]: the draws as sampler.sample() returns them, one
 * array per monitored parameter (mcmc.js:1009-1029), straight from the device into those arrays (amwg_fetch_draws_slices) */
static napi_value FetchDrawsSplit(napi_env env, napi_callback_info info) {
  napi_value a[4];
  if (!get_args(env, info, 4, a)) return NULL;
  amwg_sampler *s = unwrap(env, a[0]);
  if (!s) return NULL;
  const int64_t rows = arg_i64(env, a[1]);
  uint32_t n = 0, n2 = 0;
  if (napi_get_array_length(env, a[2], &n) != napi_ok || napi_get_array_length(env, a[3], &n2) != napi_ok || n != n2 || rows < 0) {
    napi_throw_type_error(env, NULL, "amwg_napi.fetchDrawsSplit: (handle, rows, bases[], lens[]) expected");
    return NULL;
  }
  int32_t *base = (int32_t *)calloc(n ? n : 1, sizeof(int32_t)), *len = (int32_t *)calloc(n ? n : 1, sizeof(int32_t));
  double **out = (double **)calloc(n ? n : 1, sizeof(double *));
  size_t *bytes = (size_t *)calloc(n ? n : 1, sizeof(size_t));
  napi_value result = NULL;
  int ok = base && len && out && bytes && napi_create_array_with_length(env, n, &result) == napi_ok;
  const size_t C = (size_t)amwg_num_chains(s);
  for (uint32_t k = 0; ok && k < n; k++) {
    napi_value e;
    napi_get_element(env, a[2], k, &e); base[k] = (int32_t)arg_i64(env, e);
    napi_get_element(env, a[3], k, &e); len[k] = (int32_t)arg_i64(env, e);
    if (len[k] < 0) { ok = 0; break; }
    const size_t cnt = (size_t)rows * (size_t)len[k] * C;
    napi_value arr = new_f64(env, cnt, &out[k]);
    if (!arr) { ok = 0; break; }
    bytes[k] = cnt * 8;
    napi_set_element(env, result, k, arr);
  }
  int rc = AMWG_OK;
  if (ok) rc = amwg_fetch_draws_slices(s, (int32_t)n, base, len, out, bytes);
  free(base); free(len); free(out); free(bytes);
  if (!ok) { napi_throw_error(env, NULL, "amwg_napi: cannot allocate the draws arrays"); return NULL; }
  return rc == AMWG_OK ? result : throw_amwg(env, rc);
}

/* sample(handle, n, thin) -> Float64Array [ceil(n/thin)][P][chains] */
static napi_value Sample(napi_env env, napi_callback_info info) {
  napi_value a[3];
  if (!get_args(env, info, 3, a)) return NULL;
  amwg_sampler *s = unwrap(env, a[0]);
  if (!s) return NULL;
  const int64_t nsteps = arg_i64(env, a[1]), thin = arg_i64(env, a[2]);
  if (nsteps < 0 || thin < 1) { napi_throw_range_error(env, NULL, "amwg_napi.sample: n >= 0 and thin >= 1 required"); return NULL; }
  const int64_t rows = (nsteps + thin - 1) / thin;
  const size_t n = (size_t)rows * (size_t)amwg_num_recorded(s) * (size_t)amwg_num_chains(s);
  double *data = NULL;
  napi_value out = new_f64(env, n, &data);
  if (!out) { napi_throw_error(env, NULL, "amwg_napi: cannot allocate the draws array"); return NULL; }
  int rc = amwg_sample(s, nsteps, thin, data, n * 8);
  return rc == AMWG_OK ? out : throw_amwg(env, rc);
}

static napi_value SetAdapting(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  amwg_sampler *s = unwrap(env, a[0]);
  if (!s) return NULL;
  bool flag = false;
  napi_coerce_to_bool(env, a[1], &a[1]);
  napi_get_value_bool(env, a[1], &flag);
  int rc = amwg_set_adapting(s, flag ? 1 : 0);
  return rc == AMWG_OK ? NULL : throw_amwg(env, rc);
}

static napi_value GetState(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  amwg_sampler *s = unwrap(env, a[0]);
  if (!s) return NULL;
  const size_t n = (size_t)amwg_num_components(s) * (size_t)amwg_num_chains(s);
  double *data = NULL;
  napi_value out = new_f64(env, n, &data);
  if (!out) return NULL;
  int rc = amwg_get_state(s, data, n * 8);
  return rc == AMWG_OK ? out : throw_amwg(env, rc);
}

static napi_value Info(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  amwg_sampler *s = unwrap(env, a[0]);
  if (!s) return NULL;
  const size_t n = (size_t)amwg_num_components(s) * (size_t)amwg_num_chains(s);
  double *pls, *accd, *inbd;
  int32_t *ac, *it, *bc;
  napi_value vpls = new_f64(env, n, &pls), vac = new_i32(env, n, &ac), vit = new_i32(env, n, &it), vbc = new_i32(env, n, &bc);
  napi_value vacc = new_f64(env, n, &accd), vinb = new_f64(env, n, &inbd);
  if (!vpls || !vac || !vit || !vbc || !vacc || !vinb) return NULL;
  int64_t *acc = (int64_t *)malloc(n * 8 + 8), *inb = (int64_t *)malloc(n * 8 + 8);
  int rc = amwg_info(s, pls, ac, it, bc, acc, inb);
  if (rc == AMWG_OK) for (size_t i = 0; i < n; i++) { accd[i] = (double)acc[i]; inbd[i] = (double)inb[i]; }
  free(acc);
  free(inb);
  if (rc != AMWG_OK) return throw_amwg(env, rc);
  napi_value o;
  NAPI_OK(napi_create_object(env, &o));
  napi_set_named_property(env, o, "prop_log_scale", vpls);
  napi_set_named_property(env, o, "acceptance_count", vac);
  napi_set_named_property(env, o, "iterations_since_adaption", vit);
  napi_set_named_property(env, o, "batch_count", vbc);
  napi_set_named_property(env, o, "accepts", vacc);
  napi_set_named_property(env, o, "inbounds", vinb);
  return o;
}

static napi_value Diag(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  amwg_sampler *s = unwrap(env, a[0]);
  if (!s) return NULL;
  const size_t C = (size_t)amwg_num_chains(s);
  const size_t np = (size_t)arg_i64(env, a[1]);
  double *un, *lp;
  int32_t *ord;
  napi_value vun = new_f64(env, C, &un), vlp = new_f64(env, C, &lp), vord = new_i32(env, C * np, &ord);
  if (!vun || !vlp || !vord) return NULL;
  uint64_t *u = (uint64_t *)malloc(C * 8 + 8);
  int rc = amwg_chain_diag(s, u, lp, ord);
  if (rc == AMWG_OK) for (size_t i = 0; i < C; i++) un[i] = (double)u[i];
  free(u);
  if (rc != AMWG_OK) return throw_amwg(env, rc);
  napi_value o;
  NAPI_OK(napi_create_object(env, &o));
  napi_set_named_property(env, o, "uniforms", vun);
  napi_set_named_property(env, o, "log_post", vlp);
  napi_set_named_property(env, o, "named_order", vord);
  return o;
}

static napi_value Moments(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  amwg_sampler *s = unwrap(env, a[0]);
  if (!s) return NULL;
  const size_t P = (size_t)amwg_num_recorded(s);
  double *m, *sd;
  napi_value vm = new_f64(env, P, &m), vsd = new_f64(env, P, &sd);
  if (!vm || !vsd) return NULL;
  int rc = amwg_last_sample_moments(s, m, sd);
  if (rc != AMWG_OK) return throw_amwg(env, rc);
  napi_value o;
  NAPI_OK(napi_create_object(env, &o));
  napi_set_named_property(env, o, "mean", vm);
  napi_set_named_property(env, o, "sd", vsd);
  return o;
}

/* setState(handle, Float64Array [P][chains]) */
static napi_value SetState(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  amwg_sampler *s = unwrap(env, a[0]);
  if (!s) return NULL;
  size_t n = 0;
  const double *st = (const double *)typed_data(env, a[1], napi_float64_array, &n);
  if (!st) { napi_throw_type_error(env, NULL, "amwg_napi.setState: state must be a Float64Array"); return NULL; }
  int rc = amwg_set_state(s, st, n * 8);
  return rc == AMWG_OK ? NULL : throw_amwg(env, rc);
}

/* convergence(handle) -> {rhat: Float64Array, ess: Float64Array} over the last sample() */
static napi_value Convergence(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  amwg_sampler *s = unwrap(env, a[0]);
  if (!s) return NULL;
  const size_t P = (size_t)amwg_num_recorded(s);
  double *r, *e;
  napi_value vr = new_f64(env, P, &r), ve = new_f64(env, P, &e);
  if (!vr || !ve) return NULL;
  int rc = amwg_last_sample_diagnostics(s, r, e);
  if (rc != AMWG_OK) return throw_amwg(env, rc);
  napi_value o;
  NAPI_OK(napi_create_object(env, &o));
  napi_set_named_property(env, o, "rhat", vr);
  napi_set_named_property(env, o, "ess", ve);
  return o;
}

/* quantiles(handle, Float64Array probs) -> Float64Array [P][n_probs] over the last sample() */
static napi_value Quantiles(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  amwg_sampler *s = unwrap(env, a[0]);
  if (!s) return NULL;
  size_t np = 0;
  const double *probs = (const double *)typed_data(env, a[1], napi_float64_array, &np);
  if (!probs || np < 1) { napi_throw_type_error(env, NULL, "amwg_napi.quantiles: probs must be a non-empty Float64Array"); return NULL; }
  double *q = NULL;
  napi_value out = new_f64(env, (size_t)amwg_num_recorded(s) * np, &q);
  if (!out) return NULL;
  int rc = amwg_last_sample_quantiles(s, probs, (int32_t)np, q);
  return rc == AMWG_OK ? out : throw_amwg(env, rc);
}

/* the shards of a multi-device sampler: JS array of handles -> C array (caller frees) */
static amwg_sampler **unwrap_group(napi_env env, napi_value arr, uint32_t *n) {
  *n = 0;
  bool is_arr = false;
  if (napi_is_array(env, arr, &is_arr) != napi_ok || !is_arr) { napi_throw_type_error(env, NULL, "amwg_napi: expected an array of sampler handles"); return NULL; }
  napi_get_array_length(env, arr, n);
  if (*n < 1) { napi_throw_type_error(env, NULL, "amwg_napi: empty array of sampler handles"); return NULL; }
  amwg_sampler **g = (amwg_sampler **)calloc(*n, sizeof *g);
  if (!g) { napi_throw_error(env, NULL, "amwg_napi: out of memory"); return NULL; }
  for (uint32_t i = 0; i < *n; i++) {
    napi_value e;
    napi_get_element(env, arr, i, &e);
    g[i] = unwrap(env, e);
    if (!g[i]) { free(g); return NULL; }
  }
  return g;
}

/* groupMoments([handles]) -> {mean, sd}: all shards' draws pooled (per-device reduction + RCCL all-reduce) */
static napi_value GroupMoments(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  uint32_t n = 0;
  amwg_sampler **g = unwrap_group(env, a[0], &n);
  if (!g) return NULL;
  const size_t P = (size_t)amwg_num_recorded(g[0]);
  double *m, *sd;
  napi_value vm = new_f64(env, P, &m), vsd = new_f64(env, P, &sd);
  if (!vm || !vsd) { free(g); return NULL; }
  int rc = amwg_group_moments(g, (int32_t)n, m, sd);
  free(g);
  if (rc != AMWG_OK) return throw_amwg(env, rc);
  napi_value o;
  NAPI_OK(napi_create_object(env, &o));
  napi_set_named_property(env, o, "mean", vm);
  napi_set_named_property(env, o, "sd", vsd);
  return o;
}

/* groupGatherDraws([handles], root, rows) -> {draws: Float64Array, offsets: [first element of every shard's block]}: the recorded draws of all shards, gathered
 * to the device of shard `root` inside the library (grouped ncclSend / ncclRecv; amwg_group_gather_draws) and copied to the host in ONE copy --
 * north_star's "RCCL gather at sample collection"; sampler.sample() uses it when options.gather is set */
static napi_value GroupGatherDraws(napi_env env, napi_callback_info info) {
  napi_value a[3];
  if (!get_args(env, info, 3, a)) return NULL;
  uint32_t n = 0;
  amwg_sampler **g = unwrap_group(env, a[0], &n);
  if (!g) return NULL;
  const int32_t root = (int32_t)arg_i64(env, a[1]);
  const int64_t rows = arg_i64(env, a[2]);
  size_t total = 0;
  for (uint32_t i = 0; i < n; i++) total += (size_t)rows * (size_t)amwg_num_recorded(g[i]) * (size_t)amwg_num_chains(g[i]);
  double *data = NULL;
  napi_value out = new_f64(env, total, &data);
  int64_t *offs = (int64_t *)calloc(n ? n : 1, sizeof(int64_t));
  if (!out || !offs) { free(g); free(offs); napi_throw_error(env, NULL, "amwg_napi: cannot allocate the draws array"); return NULL; }
  int rc = amwg_group_gather_draws(g, (int32_t)n, root, NULL, data, total * 8, offs);
  free(g);
  if (rc != AMWG_OK) { free(offs); return throw_amwg(env, rc); }
  napi_value o, arr;
  NAPI_OK(napi_create_object(env, &o));
  NAPI_OK(napi_create_array_with_length(env, n, &arr));
  for (uint32_t i = 0; i < n; i++) { napi_value v; NAPI_OK(napi_create_double(env, (double)offs[i], &v)); NAPI_OK(napi_set_element(env, arr, i, v)); }
  free(offs);
  napi_set_named_property(env, o, "draws", out);
  napi_set_named_property(env, o, "offsets", arr);
  return o;
}

/* groupConvergence([handles]) -> {rhat, ess} over the chains of all shards */
static napi_value GroupConvergence(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  uint32_t n = 0;
  amwg_sampler **g = unwrap_group(env, a[0], &n);
  if (!g) return NULL;
  const size_t P = (size_t)amwg_num_recorded(g[0]);
  double *r, *e;
  napi_value vr = new_f64(env, P, &r), ve = new_f64(env, P, &e);
  if (!vr || !ve) { free(g); return NULL; }
  int rc = amwg_group_diagnostics(g, (int32_t)n, r, e);
  free(g);
  if (rc != AMWG_OK) return throw_amwg(env, rc);
  napi_value o;
  NAPI_OK(napi_create_object(env, &o));
  napi_set_named_property(env, o, "rhat", vr);
  napi_set_named_property(env, o, "ess", ve);
  return o;
}

/* groupQuantiles([handles], Float64Array probs) -> Float64Array [P][n_probs] over the pooled draws of all shards */
static napi_value GroupQuantiles(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  uint32_t n = 0;
  amwg_sampler **g = unwrap_group(env, a[0], &n);
  if (!g) return NULL;
  size_t np = 0;
  const double *probs = (const double *)typed_data(env, a[1], napi_float64_array, &np);
  if (!probs || np < 1) { free(g); napi_throw_type_error(env, NULL, "amwg_napi.groupQuantiles: probs must be a non-empty Float64Array"); return NULL; }
  double *q = NULL;
  napi_value out = new_f64(env, (size_t)amwg_num_recorded(g[0]) * np, &q);
  if (!out) { free(g); return NULL; }
  int rc = amwg_group_quantiles(g, (int32_t)n, probs, (int32_t)np, q);
  free(g);
  return rc == AMWG_OK ? out : throw_amwg(env, rc);
}

/* {hits, misses, dir} of the on-disk cache of compiled closures (amwg_code_cache_stats) */
static napi_value CodeCacheStats(napi_env env, napi_callback_info info) {
  (void)info;
  int64_t h = 0, m = 0;
  char dir[1024];
  dir[0] = 0;
  int rc = amwg_code_cache_stats(&h, &m, dir, sizeof dir);
  if (rc != AMWG_OK) return throw_amwg(env, rc);
  napi_value o, t;
  NAPI_OK(napi_create_object(env, &o));
  napi_create_double(env, (double)h, &t); napi_set_named_property(env, o, "hits", t);
  napi_create_double(env, (double)m, &t); napi_set_named_property(env, o, "misses", t);
  napi_create_string_utf8(env, dir, NAPI_AUTO_LENGTH, &t); napi_set_named_property(env, o, "dir", t);
  return o;
}

static napi_value LaunchInfo(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  amwg_sampler *s = unwrap(env, a[0]);
  if (!s) return NULL;
  int32_t v[5];
  double ms = 0;
  int rc = amwg_launch_info(s, &v[0], &v[1], &v[2], &v[3], &v[4], &ms);
  if (rc != AMWG_OK) return throw_amwg(env, rc);
  static const char *names[5] = {"lanes_per_chain", "block_threads", "grid_blocks", "lds_bytes", "n_launches"};
  napi_value o, t;
  NAPI_OK(napi_create_object(env, &o));
  for (int i = 0; i < 5; i++) { napi_create_int32(env, v[i], &t); napi_set_named_property(env, o, names[i], t); }
  napi_create_double(env, ms, &t);
  napi_set_named_property(env, o, "kernel_ms", t);
  napi_create_string_utf8(env, amwg_kernel_name(s), NAPI_AUTO_LENGTH, &t);      /* the step kernel, as a profiler lists it */
  napi_set_named_property(env, o, "kernel", t);
  napi_create_int32(env, amwg_summation_order(s), &t);      /* 1: decisions and log_post in the reference's own order */
  napi_set_named_property(env, o, "summation_order", t);
  /* lanes_per_chain: -2 (AMWG_LANES_AUTOTUNE): what was timed at construction, [{lanes_per_chain, ms}, ...] */
  int32_t tl[16];
  double tm[16];
  int nt = amwg_tuning(s, tl, tm, 16);
  if (nt > 0) {
    napi_value arr;
    NAPI_OK(napi_create_array_with_length(env, (size_t)(nt < 16 ? nt : 16), &arr));
    for (int i = 0; i < nt && i < 16; i++) {
      napi_value e, x;
      napi_create_object(env, &e);
      napi_create_int32(env, tl[i], &x); napi_set_named_property(env, e, "lanes_per_chain", x);
      napi_create_double(env, tm[i], &x); napi_set_named_property(env, e, "ms", x);
      napi_set_element(env, arr, (uint32_t)i, e);
    }
    napi_set_named_property(env, o, "tuning", arr);
  }
  return o;
}

static napi_value Version(napi_env env, napi_callback_info info) {
  (void)info;
  napi_value v;
  NAPI_OK(napi_create_string_utf8(env, amwg_version(), NAPI_AUTO_LENGTH, &v));
  return v;
}

static napi_value MathExp(napi_env env, napi_callback_info info) {
  napi_value a[1], r;
  if (!get_args(env, info, 1, a)) return NULL;
  double x = 0;
  napi_get_value_double(env, a[0], &x);
  NAPI_OK(napi_create_double(env, amwg_exp(x), &r));
  return r;
}

static napi_value MathLog(napi_env env, napi_callback_info info) {
  napi_value a[1], r;
  if (!get_args(env, info, 1, a)) return NULL;
  double x = 0;
  napi_get_value_double(env, a[0], &x);
  NAPI_OK(napi_create_double(env, amwg_log(x), &r));
  return r;
}

static napi_value Uniform(napi_env env, napi_callback_info info) {
  napi_value a[3], r;
  if (!get_args(env, info, 3, a)) return NULL;
  double s = 0, c = 0, i = 0;
  napi_get_value_double(env, a[0], &s);
  napi_get_value_double(env, a[1], &c);
  napi_get_value_double(env, a[2], &i);
  NAPI_OK(napi_create_double(env, amwg_uniform((uint64_t)s, (uint64_t)c, (uint64_t)i), &r));
  return r;
}

static napi_value Init(napi_env env, napi_value exports) {
  static const struct { const char *name; napi_callback fn; } fns[] = {
      {"create", Create}, {"createUser", CreateUser}, {"compileUser", CompileUser}, {"destroy", Destroy}, {"burn", Burn}, {"burnAsync", BurnAsync}, {"sync", Sync},
      {"sample", Sample}, {"sampleAsync", SampleAsync}, {"fetchDraws", FetchDraws}, {"fetchDrawsSplit", FetchDrawsSplit}, {"setAdapting", SetAdapting},
      {"getState", GetState}, {"setState", SetState}, {"convergence", Convergence}, {"quantiles", Quantiles}, {"groupMoments", GroupMoments}, {"groupGatherDraws", GroupGatherDraws}, {"groupConvergence", GroupConvergence}, {"groupQuantiles", GroupQuantiles}, {"info", Info}, {"diag", Diag}, {"moments", Moments}, {"launchInfo", LaunchInfo}, {"codeCacheStats", CodeCacheStats},
      {"version", Version}, {"mathExp", MathExp}, {"mathLog", MathLog}, {"uniform", Uniform}};
  for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
    napi_value f;
    if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
    if (napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) return NULL;
  }
  return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
