"""ctypes binding of libamwg.so (include/amwg.h) for bench.py and the Python tests.

The product's host language is JavaScript (bayes.js_amd/mcmc.js over the N-API shim); this
binding exists because the driver's bench/test harness is Python.  It carries no logic
beyond marshalling: every computation happens in the HIP library, and loading fails loudly
if the library is missing.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AMWG_LIB") or os.path.join(HERE, "csrc", "libamwg.so")   # AMWG_LIB: development builds only
MODEL_ID = {"normal": 1, "beta_bern": 2, "hier_normal": 3, "pois_glm": 4}


class ParamDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("len", C.c_int32), ("top", C.c_int32), ("multidim", C.c_int32),
                ("lower", C.c_double), ("upper", C.c_double)]


class CompOpt(C.Structure):
    _fields_ = [("prop_log_scale", C.c_double), ("max_adaptation", C.c_double), ("initial_adaptation", C.c_double),
                ("target_accept_rate", C.c_double), ("batch_size", C.c_double), ("is_adapting", C.c_int32)]


class ModelDesc(C.Structure):
    _fields_ = [("model", C.c_int32), ("n_obs", C.c_int32), ("x", C.POINTER(C.c_double)), ("y", C.POINTER(C.c_double)),
                ("g", C.POINTER(C.c_int32)), ("G", C.c_int32), ("K", C.c_int32), ("hyper", C.c_double * 8)]


DEFAULT_HYPER = {"normal": [0, 100, 0, 100], "beta_bern": [2, 2], "hier_normal": [0, 100, 0, 100, 10], "pois_glm": [0, 10]}


class Options(C.Structure):
    _fields_ = [("chains", C.c_int64), ("seed", C.c_uint64), ("chain_offset", C.c_uint64), ("device", C.c_int32),
                ("lanes_per_chain", C.c_int32), ("block_threads", C.c_int32), ("steps_per_launch", C.c_int32),
                ("exact_division", C.c_int32), ("group_local", C.c_int32), ("full_evaluation", C.c_int32), ("test_bound_shift", C.c_int32), ("sufficient_statistics", C.c_int32)]


class UserModel(C.Structure):
    _fields_ = [("source", C.c_char_p), ("n_arrays", C.c_int32), ("arrays", C.POINTER(C.POINTER(C.c_double))),
                ("array_len", C.POINTER(C.c_int64)), ("array_type", C.POINTER(C.c_int32)), ("n_derived", C.c_int32), ("lds_bytes", C.c_int32), ("lds_bytes_one_lane", C.c_int32),
                ("parallel", C.c_int32), ("max_threads", C.c_int32), ("work_per_eval", C.c_double), ("work_one_lane", C.c_double),
                ("rows_n_obs", C.c_int32), ("rows_groups", C.c_int32), ("rows_sweep", C.c_int32)]


EXPORTS = ["amwg_kernel_name", "amwg_summation_order", "amwg_group_gather_draws", "amwg_group_comm_info", "amwg_comm_unique_id", "amwg_comm_create", "amwg_comm_info", "amwg_comm_gather_draws", "amwg_comm_moments", "amwg_comm_destroy", "amwg_code_cache_stats", "amwg_tuning", "amwg_group_moments", "amwg_group_diagnostics", "amwg_group_quantiles", "amwg_last_sample_quantiles", "amwg_fp64_peak", "amwg_set_state", "amwg_last_sample_diagnostics", "amwg_create_user", "amwg_compile_user", "amwg_num_recorded", "amwg_create", "amwg_burn", "amwg_burn_async", "amwg_sample", "amwg_sample_async", "amwg_fetch_draws", "amwg_fetch_draws_slices", "amwg_sample_device", "amwg_set_adapting", "amwg_get_state", "amwg_info", "amwg_chain_diag", "amwg_last_sample_moments", "amwg_sync", "amwg_num_components", "amwg_num_chains", "amwg_launch_info", "amwg_destroy", "amwg_last_error", "amwg_version", "amwg_exp", "amwg_log", "amwg_uniform"]      # include/amwg.h: the product library
SELFTEST_EXPORTS = ["amwg_prefault_selftest", "amwg_math1", "amwg_math2", "amwg_hypot3", "amwg_log1p", "amwg_expm1", "amwg_two_valued_sum_check", "amwg_pow", "amwg_ld_host", "amwg_ld_device", "amwg_device_eval"]      # include/amwg_selftest.h: libamwg_selftest.so only

_lib = None


def lib():
    """Loads libamwg.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libamwg.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(LIB_PATH)
        vp, i32, i64, u64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double
        pd, pi32, pi64, pu64 = C.POINTER(dbl), C.POINTER(i32), C.POINTER(i64), C.POINTER(u64)
        L.amwg_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(ParamDesc), i32, pd, C.POINTER(CompOpt),
                                  C.POINTER(Options), C.POINTER(vp)]
        L.amwg_burn.argtypes = [vp, i64]
        L.amwg_burn_async.argtypes = [vp, i64]
        L.amwg_sample_async.argtypes = [vp, i64, i64]
        L.amwg_fetch_draws.argtypes = [vp, pd, C.c_size_t]
        L.amwg_fetch_draws_slices.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_size_t)]
        L.amwg_sample.argtypes = [vp, i64, i64, pd, C.c_size_t]
        L.amwg_sample_device.argtypes = [vp, i64, i64, vp, C.c_size_t]
        L.amwg_set_adapting.argtypes = [vp, i32]
        L.amwg_get_state.argtypes = [vp, pd, C.c_size_t]
        L.amwg_info.argtypes = [vp, pd, pi32, pi32, pi32, pi64, pi64]
        L.amwg_chain_diag.argtypes = [vp, pu64, pd, pi32]
        L.amwg_last_sample_moments.argtypes = [vp, pd, pd]
        L.amwg_sync.argtypes = [vp]
        L.amwg_num_components.argtypes = [vp]
        L.amwg_num_chains.argtypes = [vp]
        L.amwg_num_chains.restype = i64
        L.amwg_launch_info.argtypes = [vp, pi32, pi32, pi32, pi32, pi32, pd]
        L.amwg_kernel_name.argtypes = [vp]
        L.amwg_summation_order.argtypes = [vp]
        L.amwg_kernel_name.restype = C.c_char_p
        L.amwg_destroy.argtypes = [vp]
        L.amwg_last_error.restype = C.c_char_p
        L.amwg_version.restype = C.c_char_p
        L.amwg_exp.restype = dbl
        L.amwg_exp.argtypes = [dbl]
        L.amwg_log.restype = dbl
        L.amwg_log.argtypes = [dbl]
        L.amwg_uniform.restype = dbl
        L.amwg_uniform.argtypes = [u64, u64, u64]
        L.amwg_compile_user.argtypes = [C.c_char_p, i32, i32, C.c_char_p, C.POINTER(C.c_size_t)]
        L.amwg_code_cache_stats.argtypes = [pi64, pi64, C.c_char_p, C.c_size_t]
        L.amwg_create_user.argtypes = [C.POINTER(UserModel), C.POINTER(ParamDesc), i32, pd, C.POINTER(CompOpt),
                                       C.POINTER(Options), C.POINTER(vp)]
        L.amwg_num_recorded.argtypes = [vp]
        L.amwg_set_state.argtypes = [vp, pd, C.c_size_t]
        L.amwg_fp64_peak.argtypes = [i32, pd]
        L.amwg_last_sample_quantiles.argtypes = [vp, pd, i32, pd]
        L.amwg_last_sample_diagnostics.argtypes = [vp, pd, pd]
        pvp = C.POINTER(vp)
        L.amwg_group_moments.argtypes = [pvp, i32, pd, pd]
        L.amwg_group_diagnostics.argtypes = [pvp, i32, pd, pd]
        L.amwg_group_quantiles.argtypes = [pvp, i32, pd, i32, pd]
        L.amwg_group_gather_draws.argtypes = [pvp, i32, i32, vp, pd, C.c_size_t, pi64]
        L.amwg_group_comm_info.argtypes = [pvp, i32, pi32, pi32, i32]
        L.amwg_comm_unique_id.argtypes = [C.c_char_p, C.c_size_t]
        L.amwg_comm_create.argtypes = [C.c_char_p, C.c_size_t, i32, i32, i32, pvp]
        L.amwg_comm_info.argtypes = [vp, pi32, pi32, pi32]
        L.amwg_comm_gather_draws.argtypes = [vp, vp, i32, vp, C.c_size_t, pi64]
        L.amwg_comm_moments.argtypes = [vp, vp, pd, pd]
        L.amwg_comm_destroy.argtypes = [vp]
        _lib = L
    return _lib


_selftest = None


def selftest_lib():
    """libamwg_selftest.so: the product's sources built with -DAMWG_SELFTEST, which adds the entry points of include/amwg_selftest.h (the
    arithmetic building blocks one by one).  Test suite only; the product library does not carry them."""
    global _selftest
    if _selftest is None:
        path = os.path.join(os.path.dirname(LIB_PATH), "libamwg_selftest.so")
        if not os.path.exists(path):
            raise RuntimeError("libamwg_selftest.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(path)
        i32, i64, dbl = C.c_int32, C.c_int64, C.c_double
        pd = C.POINTER(dbl)
        L.amwg_last_error.restype = C.c_char_p
        L.amwg_device_eval.argtypes = [i32, i32, i64, pd, pd, pd, pd]
        L.amwg_two_valued_sum_check.argtypes = [i32, pd, i32, i64, pd, pd, pd, pd, pd]
        L.amwg_pow.restype = dbl
        L.amwg_pow.argtypes = [dbl, dbl]
        L.amwg_math1.restype = dbl
        L.amwg_math1.argtypes = [i32, dbl]
        L.amwg_math2.restype = dbl
        L.amwg_math2.argtypes = [i32, dbl, dbl]
        L.amwg_hypot3.restype = dbl
        L.amwg_hypot3.argtypes = [dbl, dbl, dbl]
        for f in ("amwg_log1p", "amwg_expm1"):
            getattr(L, f).restype = dbl
            getattr(L, f).argtypes = [dbl]
        L.amwg_ld_host.restype = dbl
        L.amwg_ld_host.argtypes = [i32, dbl, dbl, dbl, dbl]
        L.amwg_ld_device.argtypes = [i32, i64, pd, pd]
        L.amwg_prefault_selftest.argtypes = [C.c_void_p, C.c_size_t, i32, i32]
        _selftest = L
    return _selftest


class AmwgError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise AmwgError("amwg error %d: %s" % (rc, lib().amwg_last_error().decode()))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Sampler:
    """Many-chain sampler handle.  `spec` = dict(model, n_obs, data{x[,y,g,G,K]}, params[], P, init[], comp_opts[]) for a
    built-in family, or dict(user={source, arrays[], n_derived, lds_bytes, parallel, max_threads}, params[], P, init[],
    comp_opts[]) for a closure translated by bayes.js_amd/translate.js (amwg_create_user)."""

    def __init__(self, spec, chains, seed, chain_offset=0, device=0, lanes_per_chain=0, block_threads=0,
                 steps_per_launch=0, exact_division=0, group_local=0, full_evaluation=0, test_bound_shift=0, sufficient_statistics=0):
        L = lib()
        keep = []
        user = spec.get("user")
        if user is None:
            d = spec["data"]
            md = ModelDesc()
            md.model = MODEL_ID[spec["model"]]
            md.n_obs = spec["n_obs"]
            x = np.ascontiguousarray(d["x"], dtype=np.float64)
            keep.append(x)
            md.x = _dp(x)
            if "y" in d:
                y = np.ascontiguousarray(d["y"], dtype=np.float64)
                keep.append(y)
                md.y = _dp(y)
            if "g" in d:
                g = np.ascontiguousarray(d["g"], dtype=np.int32)
                keep.append(g)
                md.g = g.ctypes.data_as(C.POINTER(C.c_int32))
            md.G = int(spec.get("G", 0))
            md.K = int(spec.get("K", 0))
            for i, v in enumerate(spec.get("hyper") or DEFAULT_HYPER[spec["model"]]):
                md.hyper[i] = float(v)
        else:
            um = UserModel()
            src = user["source"].encode() if isinstance(user["source"], str) else user["source"]
            keep.append(src)
            um.source = src
            arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in user["arrays"]]
            keep.append(arrs)
            um.n_arrays = len(arrs)
            ptrs = (C.POINTER(C.c_double) * max(1, len(arrs)))(*[_dp(a) for a in arrs])
            lens = (C.c_int64 * max(1, len(arrs)))(*[a.size for a in arrs])
            types = (C.c_int32 * max(1, len(arrs)))(*[int(t) for t in user.get("array_types", [0] * len(arrs))])
            keep += [ptrs, lens, types]
            um.arrays, um.array_len, um.array_type = ptrs, lens, types
            um.n_derived, um.lds_bytes = int(user.get("n_derived", 0)), int(user.get("lds_bytes", 0))
            um.lds_bytes_one_lane = int(user.get("lds_bytes_one_lane", 0))
            um.parallel, um.max_threads = int(user.get("parallel", 0)), int(user.get("max_threads", 0))
            um.work_per_eval = float(user.get("work_per_eval", 0.0))
            um.work_one_lane = float(user.get("work_one_lane", 0.0))
            um.rows_n_obs, um.rows_groups, um.rows_sweep = int(user.get("rows_n_obs", 0)), int(user.get("rows_groups", 0)), int(user.get("rows_sweep", 0))
        n = len(spec["params"])
        pa = (ParamDesc * n)()
        TYPE = {"real": 0, "int": 1, "binary": 2}
        for i, p in enumerate(spec["params"]):
            pa[i].type = TYPE[p["type"]]
            pa[i].len, pa[i].top, pa[i].multidim = p["len"], p["top"], p["multidim"]
            pa[i].lower, pa[i].upper = p["lower"], p["upper"]
        P = spec["P"]
        oa = (CompOpt * P)()
        for i, o in enumerate(spec["comp_opts"]):
            oa[i].prop_log_scale = o["prop_log_scale"]
            oa[i].max_adaptation = o["max_adaptation"]
            oa[i].initial_adaptation = o["initial_adaptation"]
            oa[i].target_accept_rate = o["target_accept_rate"]
            oa[i].batch_size = float(o["batch_size"])
            oa[i].is_adapting = int(bool(o["is_adapting"]))
        init = np.ascontiguousarray(spec["init"], dtype=np.float64)
        op = Options()
        op.chains, op.seed, op.chain_offset, op.device = chains, seed, chain_offset, device
        op.lanes_per_chain, op.block_threads, op.steps_per_launch = lanes_per_chain, block_threads, steps_per_launch
        op.exact_division = exact_division
        op.group_local = group_local
        op.full_evaluation = full_evaluation
        op.test_bound_shift = test_bound_shift
        op.sufficient_statistics = sufficient_statistics
        h = C.c_void_p()
        if user is None:
            _check(L.amwg_create(C.byref(md), pa, n, _dp(init), oa, C.byref(op), C.byref(h)))
        else:
            _check(L.amwg_create_user(C.byref(um), pa, n, _dp(init), oa, C.byref(op), C.byref(h)))
        self.h = h
        self.P = P
        self.PR = L.amwg_num_recorded(h)   # values per draw row: P + derived quantities
        self.C = chains
        self.n_params = n

    def close(self):
        if getattr(self, "h", None):
            lib().amwg_destroy(self.h)
            self.h = None

    __del__ = close

    def burn(self, n):
        _check(lib().amwg_burn(self.h, n))

    def sample(self, n, thin=1):
        """-> array [kept][P + derived][chains]"""
        kept = -(-n // thin)
        out = np.empty((kept, self.PR, self.C), dtype=np.float64)
        _check(lib().amwg_sample(self.h, n, thin, _dp(out), out.nbytes))
        return out

    def burn_async(self, n):
        _check(lib().amwg_burn_async(self.h, n))

    def sample_async(self, n, thin=1):
        _check(lib().amwg_sample_async(self.h, n, thin))
        self._pending = -(-n // thin)

    def fetch_draws(self):
        out = np.empty((self._pending, self.PR, self.C), dtype=np.float64)
        _check(lib().amwg_fetch_draws(self.h, _dp(out), out.nbytes))
        return out

    def fetch_draws_slices(self, slices):
        """slices: [(base, len), ...] -> one array [kept][len][chains] per slice (amwg_fetch_draws_slices)"""
        outs = [np.empty((self._pending, ln, self.C), dtype=np.float64) for _, ln in slices]
        n = len(slices)
        base = (C.c_int32 * max(n, 1))(*[b for b, _ in slices])
        ln = (C.c_int32 * max(n, 1))(*[l for _, l in slices])
        ptrs = (C.POINTER(C.c_double) * max(n, 1))(*[_dp(o) for o in outs])
        sizes = (C.c_size_t * max(n, 1))(*[o.nbytes for o in outs])
        _check(lib().amwg_fetch_draws_slices(self.h, n, base, ln, ptrs, sizes))
        return outs

    def sample_device(self, n, thin, dev_ptr, nbytes):
        _check(lib().amwg_sample_device(self.h, n, thin, C.c_void_p(dev_ptr), nbytes))

    def sync(self):
        _check(lib().amwg_sync(self.h))

    def set_adapting(self, flag):
        _check(lib().amwg_set_adapting(self.h, int(bool(flag))))

    def state(self):
        out = np.empty((self.P, self.C), dtype=np.float64)
        _check(lib().amwg_get_state(self.h, _dp(out), out.nbytes))
        return out

    def info(self):
        P, Cn = self.P, self.C
        pls = np.empty((P, Cn))
        ac, it, bc = (np.empty((P, Cn), dtype=np.int32) for _ in range(3))
        acc, inb = (np.empty((P, Cn), dtype=np.int64) for _ in range(2))
        i32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        i64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
        _check(lib().amwg_info(self.h, _dp(pls), i32(ac), i32(it), i32(bc), i64(acc), i64(inb)))
        return {"prop_log_scale": pls, "acceptance_count": ac, "iterations_since_adaption": it, "batch_count": bc,
                "accepts": acc, "inbounds": inb}

    def diag(self):
        un = np.empty(self.C, dtype=np.uint64)
        lp = np.empty(self.C)
        order = np.empty((self.C, self.n_params), dtype=np.int32)
        _check(lib().amwg_chain_diag(self.h, un.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(lp),
                                     order.ctypes.data_as(C.POINTER(C.c_int32))))
        return {"uniforms": un, "log_post": lp, "named_order": order}

    def moments(self):
        m, s = np.empty(self.PR), np.empty(self.PR)
        _check(lib().amwg_last_sample_moments(self.h, _dp(m), _dp(s)))
        return m, s

    def set_state(self, state):
        st = np.ascontiguousarray(state, dtype=np.float64)
        assert st.shape == (self.P, self.C)
        _check(lib().amwg_set_state(self.h, _dp(st), st.nbytes))

    def convergence(self):
        r, e = np.empty(self.PR), np.empty(self.PR)
        _check(lib().amwg_last_sample_diagnostics(self.h, _dp(r), _dp(e)))
        return r, e

    def quantiles(self, probs):
        """-> array [P + derived][len(probs)] over all chains x kept draws of the last sample()"""
        pr = np.ascontiguousarray(probs, dtype=np.float64)
        out = np.empty((self.PR, pr.size))
        _check(lib().amwg_last_sample_quantiles(self.h, _dp(pr), pr.size, _dp(out)))
        return out

    def tuning(self):
        """AMWG_LANES_AUTOTUNE (lanes_per_chain=-2): [(lanes per chain, ms of the timing run)] of every candidate timed at construction"""
        L = lib()
        L.amwg_tuning.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_int32]
        lanes, ms = (C.c_int32 * 16)(), (C.c_double * 16)()
        n = L.amwg_tuning(self.h, lanes, ms, 16)
        return [(lanes[i], ms[i]) for i in range(min(n, 16))]

    def launch_info(self):
        v = [C.c_int32() for _ in range(5)]
        ms = C.c_double()
        _check(lib().amwg_launch_info(self.h, *[C.byref(x) for x in v], C.byref(ms)))
        return {"lanes_per_chain": v[0].value, "block_threads": v[1].value, "grid_blocks": v[2].value,
                "lds_bytes": v[3].value, "n_launches": v[4].value, "kernel_ms": ms.value, "kernel": (lib().amwg_kernel_name(self.h) or b"").decode(),
                "summation_order": lib().amwg_summation_order(self.h)}


def _group(samplers):
    arr = (C.c_void_p * len(samplers))(*[s.h for s in samplers])
    return arr, len(samplers), samplers[0].PR


def group_moments(samplers):
    """mean, sd over the pooled draws of several samplers (the shards of one job): amwg_group_moments (RCCL all-reduce)"""
    arr, n, PR = _group(samplers)
    m, sd = np.empty(PR), np.empty(PR)
    _check(lib().amwg_group_moments(arr, n, _dp(m), _dp(sd)))
    return m, sd


def group_convergence(samplers):
    arr, n, PR = _group(samplers)
    r, e = np.empty(PR), np.empty(PR)
    _check(lib().amwg_group_diagnostics(arr, n, _dp(r), _dp(e)))
    return r, e


def group_quantiles(samplers, probs):
    arr, n, PR = _group(samplers)
    pr = np.ascontiguousarray(probs, dtype=np.float64)
    out = np.empty((PR, pr.size))
    _check(lib().amwg_group_quantiles(arr, n, _dp(pr), pr.size, _dp(out)))
    return out


def group_gather_draws(samplers, root=0, dst_device=None, to_host=True):
    """amwg_group_gather_draws: the recorded draws of every shard on the device of shard `root`, blocks back to back in shard order
    -> (list of per-shard arrays [kept][P + derived][chains_i] if to_host else None, offsets)"""
    arr, n, PR = _group(samplers)
    rows = samplers[0]._pending
    sizes = [rows * PR * s.C for s in samplers]
    total = sum(sizes)
    host = np.empty(total, dtype=np.float64) if to_host else None
    offs = (C.c_int64 * n)()
    _check(lib().amwg_group_gather_draws(arr, n, root, C.c_void_p(dst_device) if dst_device else None, _dp(host) if to_host else None, total * 8, offs))
    blocks = None
    if to_host:
        blocks = [host[offs[i]:offs[i] + sizes[i]].reshape(rows, PR, samplers[i].C) for i in range(n)]
    return blocks, list(offs)


def group_comm_info(samplers):
    arr, n, _ = _group(samplers)
    nr = C.c_int32()
    devs = (C.c_int32 * max(n, 1))()
    _check(lib().amwg_group_comm_info(arr, n, C.byref(nr), devs, n))
    return {"rccl_ranks_seen": nr.value, "devices": [devs[i] for i in range(min(n, nr.value))]}


class Comm:
    """A communicator over the processes of a one-process-per-device job (amwg_comm_*).  `exchange(id_bytes_or_None) -> id_bytes` carries rank 0's
    128-byte id to every rank (e.g. a torch.distributed broadcast)."""

    def __init__(self, n_ranks, rank, device, exchange):
        buf = C.create_string_buffer(128)
        if rank == 0:
            _check(lib().amwg_comm_unique_id(buf, 128))
        ident = exchange(buf.raw if rank == 0 else None)
        h = C.c_void_p()
        _check(lib().amwg_comm_create(ident, len(ident), n_ranks, rank, device, C.byref(h)))
        self.h, self.n_ranks, self.rank = h, n_ranks, rank

    def info(self):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        _check(lib().amwg_comm_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"rccl_ranks_seen": a.value, "rank": b.value, "device": c.value}

    def gather_draws(self, sampler, root, dst_device_ptr, capacity_bytes):
        counts = (C.c_int64 * self.n_ranks)()
        _check(lib().amwg_comm_gather_draws(sampler.h, self.h, root, C.c_void_p(dst_device_ptr) if dst_device_ptr else None, capacity_bytes, counts))
        return list(counts)

    def moments(self, sampler):
        m, s = np.empty(sampler.PR), np.empty(sampler.PR)
        _check(lib().amwg_comm_moments(sampler.h, self.h, _dp(m), _dp(s)))
        return m, s

    def close(self):
        if getattr(self, "h", None):
            lib().amwg_comm_destroy(self.h)
            self.h = None


def code_cache_stats():
    """(hits, misses, directory) of the on-disk cache of compiled closures in this process."""
    h, m = C.c_int64(0), C.c_int64(0)
    buf = C.create_string_buffer(1024)
    _check(lib().amwg_code_cache_stats(C.byref(h), C.byref(m), buf, 1024))
    return h.value, m.value, buf.value.decode()


def fp64_peak(device=0):
    """Measured fp64 fma issue rate of the device, lane-operations per second."""
    v = C.c_double()
    _check(lib().amwg_fp64_peak(device, C.byref(v)))
    return v.value


def device_eval(op, a, b=None, c=None, device=0):
    a = np.ascontiguousarray(a, dtype=np.float64)
    out = np.empty_like(a)
    bb = np.ascontiguousarray(b, dtype=np.float64) if b is not None else None
    cc = np.ascontiguousarray(c, dtype=np.float64) if c is not None else None
    _check(selftest_lib().amwg_device_eval(device, op, a.size, _dp(a), _dp(bb) if bb is not None else None,
                                  _dp(cc) if cc is not None else None, _dp(out)))
    return out
