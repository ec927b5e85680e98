"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE: the CPU checker)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODEL_ID = {"normal": 1, "beta_bern": 2, "hier_normal": 3, "pois_glm": 4}


class OrcParam(C.Structure):
    _fields_ = [("type", C.c_int32), ("len", C.c_int32), ("top", C.c_int32), ("multidim", C.c_int32),
                ("lower", C.c_double), ("upper", C.c_double)]


class OrcCompOpt(C.Structure):
    _fields_ = [("prop_log_scale", C.c_double), ("max_adaptation", C.c_double), ("initial_adaptation", C.c_double),
                ("target_accept_rate", C.c_double), ("batch_size", C.c_double), ("is_adapting", C.c_int32)]


class OrcData(C.Structure):
    _fields_ = [("model", C.c_int32), ("n_obs", C.c_int32), ("x", C.POINTER(C.c_double)), ("y", C.POINTER(C.c_double)),
                ("g", C.POINTER(C.c_int32)), ("G", C.c_int32), ("K", C.c_int32), ("hyper", C.c_double * 8)]


DEFAULT_HYPER = {"normal": [0, 100, 0, 100], "beta_bern": [2, 2], "hier_normal": [0, 100, 0, 100, 10], "pois_glm": [0, 10]}


LOG_POST_FN = C.CFUNCTYPE(C.c_double, C.POINTER(C.c_double), C.c_int, C.c_void_p)
TYPE_ID = {"real": 0, "int": 1, "binary": 2}
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(OrcData), C.POINTER(OrcParam), C.c_int, C.POINTER(C.c_double),
                                 C.POINTER(OrcCompOpt), C.c_uint64, C.c_uint64, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_num_components.argtypes = [C.c_void_p]
        L.orc_burn.argtypes = [C.c_void_p, C.c_int64]
        L.orc_sample.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_double)]
        L.orc_set_adapting.argtypes = [C.c_void_p, C.c_int]
        L.orc_get_state.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.orc_get_info.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.orc_uniforms_used.restype = C.c_uint64
        L.orc_uniforms_used.argtypes = [C.c_void_p]
        L.orc_log_post.restype = C.c_double
        L.orc_log_post.argtypes = [C.c_void_p]
        L.orc_log_post_unhoisted.restype = C.c_double
        L.orc_log_post_unhoisted.argtypes = [C.c_void_p]
        L.orc_named_order.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.orc_uniform.restype = C.c_double
        L.orc_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        for f in ("orc_exp", "orc_log", "orc_js_round", "orc_lgamma"):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = [C.c_double]
        for f, n in (("orc_ld_norm", 3), ("orc_ld_unif", 3), ("orc_ld_beta", 3), ("orc_ld_bern", 2), ("orc_ld_pois", 2)):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = [C.c_double] * n
        L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_set_callback.argtypes = [LOG_POST_FN, C.c_void_p]
        for f in ("orc_log1p", "orc_expm1", "orc_tanh", "orc_atan", "orc_log10"):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = [C.c_double]
        L.orc_pow.restype = C.c_double
        L.orc_pow.argtypes = [C.c_double, C.c_double]
        L.orc_ld.restype = C.c_double
        L.orc_ld.argtypes = [C.c_int] + [C.c_double] * 4
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class OracleChain:
    """One reference-order chain.  `spec` is a dict from model_spec.build_spec(), or -- for a user closure -- a dict with
    `log_post_fn(state ndarray[P], lanes) -> float` instead of model/data (the oracle steps, the callback evaluates)."""

    def __init__(self, spec, seed, chain, lanes=1, group_local=False):
        L = lib()
        self.spec = spec
        self._keep = []
        od = OrcData()
        if "log_post_fn" in spec:
            fn, P_ = spec["log_post_fn"], spec["P"]
            self._cb = LOG_POST_FN(lambda st, ln, ctx: float(fn(np.ctypeslib.as_array(st, shape=(P_,)).copy(), ln)))
            L.orc_set_callback(self._cb, None)
            od.model = 5   # ORC_MODEL_CALLBACK
        else:
            d = spec["data"]
            od.model = MODEL_ID[spec["model"]]
            od.n_obs = spec["n_obs"]
            x = np.ascontiguousarray(d["x"], dtype=np.float64)
            self._keep.append(x)
            od.x = _dp(x)
            if "y" in d:
                y = np.ascontiguousarray(d["y"], dtype=np.float64)
                self._keep.append(y)
                od.y = _dp(y)
            if "g" in d:
                g = np.ascontiguousarray(d["g"], dtype=np.int32)
                self._keep.append(g)
                od.g = g.ctypes.data_as(C.POINTER(C.c_int32))
            od.G = spec.get("G", 0)
            od.K = spec.get("K", 0)
            for i, v in enumerate(spec.get("hyper") or DEFAULT_HYPER[spec["model"]]):
                od.hyper[i] = float(v)
        n = len(spec["params"])
        pa = (OrcParam * n)()
        for i, p in enumerate(spec["params"]):
            pa[i].type = TYPE_ID[p["type"]]
            pa[i].len = p["len"]
            pa[i].top = p["top"]
            pa[i].multidim = p["multidim"]
            pa[i].lower = p["lower"]
            pa[i].upper = p["upper"]
        P = spec["P"]
        self.P = P
        oa = (OrcCompOpt * P)()
        for i, o in enumerate(spec["comp_opts"]):
            oa[i].prop_log_scale = o["prop_log_scale"]
            oa[i].max_adaptation = o["max_adaptation"]
            oa[i].initial_adaptation = o["initial_adaptation"]
            oa[i].target_accept_rate = o["target_accept_rate"]
            oa[i].batch_size = float(o["batch_size"])
            oa[i].is_adapting = int(bool(o["is_adapting"]))
        init = np.ascontiguousarray(spec["init"], dtype=np.float64)
        self.n_params = n
        self.h = L.orc_create(C.byref(od), pa, n, _dp(init), oa, seed, chain, lanes)
        assert self.h
        if group_local:
            L.orc_set_group_local.argtypes = [C.c_void_p, C.c_int]
            assert L.orc_set_group_local(self.h, 1) == 0, "group-local evaluation: preconditions not met"

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_destroy(self.h)
            self.h = None

    def burn(self, n):
        lib().orc_burn(self.h, n)

    def sample(self, n, thin=1):
        kept = -(-n // thin)
        out = np.empty((kept, self.P), dtype=np.float64)
        lib().orc_sample(self.h, n, thin, _dp(out))
        return out

    def set_adapting(self, flag):
        lib().orc_set_adapting(self.h, int(flag))

    def state(self):
        out = np.empty(self.P)
        lib().orc_get_state(self.h, _dp(out))
        return out

    def info(self):
        P = self.P
        pls = np.empty(P)
        ac, it, bc = (np.empty(P, dtype=np.int32) for _ in range(3))
        acc, inb = (np.empty(P, dtype=np.int64) for _ in range(2))
        i32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        i64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
        lib().orc_get_info(self.h, _dp(pls), i32(ac), i32(it), i32(bc), i64(acc), i64(inb))
        return {"prop_log_scale": pls, "acceptance_count": ac, "iterations_since_adaption": it, "batch_count": bc,
                "accepts": acc, "inbounds": inb}

    def uniforms(self):
        return int(lib().orc_uniforms_used(self.h))

    def log_post(self):
        return float(lib().orc_log_post(self.h))

    def log_post_unhoisted(self):
        return float(lib().orc_log_post_unhoisted(self.h))

    def named_order(self):
        o = np.empty(self.n_params, dtype=np.int32)
        lib().orc_named_order(self.h, o.ctypes.data_as(C.POINTER(C.c_int32)))
        return o
