"""Builds the flat model description shared by the oracle binding and the HIP library binding.

The description is what the reference sampler holds after complete_params()
(mcmc.js:357-403) and stepper construction (mcmc.js:500-505, 644-649, 869-878): tests
take it either from a golden fixture (as the reference built it) or from defaults().
"""
import numpy as np

import synth

DEFAULT_OPT = {"prop_log_scale": 0.0, "batch_size": 50, "max_adaptation": 0.33, "initial_adaptation": 1.0,
               "target_accept_rate": 0.44, "is_adapting": True}   # mcmc.js:500-505
INF = float("inf")


def make_data(model, n_obs, data_seed, G=32, exp=None):
    if model == "normal":
        return synth.normal(n_obs, data_seed)
    if model == "beta_bern":
        return synth.bern(n_obs, data_seed)
    if model == "hier_normal":
        d = synth.hier(n_obs, G, data_seed)
        return {"x": d["y"], "g": d["g"], "G": G}
    if model == "pois_glm":
        d = synth.glm(n_obs, data_seed, exp=exp)
        return {"x": d["X"].reshape(-1), "y": d["y"], "K": d["K"]}
    raise ValueError(model)


def default_params(model, n_obs, G=32):
    """Completed params of oracle/ref_models.js, in Object.keys order."""
    def P(type="real", dim=(1,), lower=-INF, upper=INF, init=None):
        ln = int(np.prod(dim))
        if init is None:   # param_init_fixed, mcmc.js:313-341
            if type == "real":
                init = 0.5 if (lower == -INF and upper == INF) else upper - 0.5 if lower == -INF else \
                    lower + 0.5 if upper == INF else (lower + upper) / 2
            else:
                init = 1 if (lower == -INF and upper == INF) else upper - 1 if lower == -INF else \
                    lower + 1 if upper == INF else float(np.floor((lower + upper) / 2 + 0.5))
        return {"type": type, "len": ln, "top": int(dim[0]), "multidim": int(tuple(dim) != (1,)), "lower": lower,
                "upper": upper, "init": [float(init)] * ln}
    if model == "normal":
        return [P(), P(lower=0.0)]
    if model == "beta_bern":
        return [P(lower=0.0, upper=1.0)]
    if model == "hier_normal":
        return [P(dim=(G,)), P(), P(lower=0.0, init=1.0)]
    if model == "pois_glm":
        return [P(dim=(8,), init=0.0), P(type="int", lower=0.0, upper=float(n_obs - 1))]
    raise ValueError(model)


def build_spec(model, data, params=None, comp_opts=None, G=None, hyper=None):
    n_obs = len(data["y"]) if model == "pois_glm" else len(data["x"])
    G = G if G is not None else data.get("G", 0)
    if params is None:
        params = default_params(model, n_obs, G or 32)
    P = sum(p["len"] for p in params)
    init = [v for p in params for v in p["init"]]
    if comp_opts is None:
        comp_opts = [dict(DEFAULT_OPT) for _ in range(P)]
    return {"model": model, "n_obs": n_obs, "data": data, "params": params, "P": P, "init": init,
            "comp_opts": comp_opts, "G": int(G or 0), "K": int(data.get("K", 0)), "hyper": hyper}


def spec_from_golden(gold, chain_rec=None):
    """Spec exactly as the reference built it for a golden case."""
    c = gold["case"]
    rec = chain_rec or gold["chains"][0]
    if "data" in gold:                     # stored (small) data
        d = gold["data"]
        if c["model"] in ("normal", "beta_bern"):
            data = {"x": np.array(d["x"], dtype=np.float64)}
        elif c["model"] == "hier_normal":
            data = {"x": np.array(d["y"], dtype=np.float64), "g": np.array(d["g"], dtype=np.int32), "G": d["G"]}
        else:
            data = {"x": np.array(d["X"], dtype=np.float64), "y": np.array(d["y"], dtype=np.float64), "K": d["K"]}
    else:
        import oracle_lib
        data = make_data(c["model"], c["N"], c["data_seed"], G=c.get("G", 32), exp=oracle_lib.lib().orc_exp)
    params = []
    for p in rec["params_completed"]:
        dim = p["dim"]
        params.append({"type": p["type"], "len": int(np.prod(dim)), "top": int(dim[0]), "multidim": int(list(dim) != [1]),
                       "lower": float(p["lower"]), "upper": float(p["upper"]), "init": [float(v) for v in p["init"]]})
    return build_spec(c["model"], data, params=params, comp_opts=rec["comp_opts"], G=data.get("G"), hyper=c.get("hyper"))
