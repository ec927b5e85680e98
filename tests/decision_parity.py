"""How often does a summation order other than the reference's change an accept decision?  (test helper; GPU only)

With one lane per chain the kernel adds the observations' terms in the reference's order and every double is the reference's (the goldens,
the live-reference test).  With G lanes per chain -- and in the opt-in group-local mode -- log_post differs from the reference's value by a
few ulps (~1e-12 relative), so `Math.exp(prop - curr) > Math.random()` (mcmc.js:527-528) decides differently only when the uniform falls
inside that sliver around exp(delta).  A flipped decision changes the chain's state, hence everything after it: a chain whose final state,
accept / in-bounds counts and uniform count all equal those of the SAME chain id run with one lane per chain had no flipped decision.

compare() runs the same seeded job in two geometries on the device and counts the chains that differ:
    decisions        in-bounds proposals evaluated (= accept tests performed) by the one-lane run, all chains
    chains_differing chains for which anything differs at the end
    first_flips      = chains_differing (every differing chain had at least one flip; after its first flip a chain is a different run, so its
                       later decisions are not comparisons of the same proposal any more)
    flips_per_1e9    first_flips / decisions_until_flip * 1e9, where the decisions of a differing chain are counted in full (a lower bound
                       on the denominator would only raise the rate by chains_differing / chains)
    lp_abs_diff      |log_post_G - log_post_1| over the agreeing chains at the end of the run (max, mean): the size of the sliver.  Since
                       |exp(a) - exp(b)| <= |a - b| for a, b <= 0, the probability that one decision flips is at most 2 x (that difference),
                       which is the analytic expectation the measured count is compared with.
"""
import numpy as np


def run_one(A, spec, chains, steps, seed, kw, steps_per_launch=0):
    """-> what compare() looks at, of one geometry (kept by tools/flip_rate.py to compare several geometries with ONE reference run)"""
    s = A.Sampler(spec, chains=chains, seed=seed, steps_per_launch=steps_per_launch, **kw)
    li = s.launch_info()
    s.burn(steps)
    out = (li, s.info(), s.diag(), s.state())
    s.close()
    return out


def compare(A, spec, chains, steps, seed, alt, ref=None, steps_per_launch=0, ref_run=None):
    ref = dict(ref or {"lanes_per_chain": 1})
    la, ia, da, sa = ref_run if ref_run is not None else run_one(A, spec, chains, steps, seed, ref, steps_per_launch)
    lb, ib, db, sb = run_one(A, spec, chains, steps, seed, alt, steps_per_launch)
    same = np.all(ia["accepts"] == ib["accepts"], axis=0) & np.all(ia["inbounds"] == ib["inbounds"], axis=0)
    same &= da["uniforms"] == db["uniforms"]
    same &= np.all(sa.view(np.uint64) == sb.view(np.uint64), axis=0)
    same &= np.all(ia["prop_log_scale"].view(np.uint64) == ib["prop_log_scale"].view(np.uint64), axis=0)
    decisions = int(ia["inbounds"].sum())
    differing = int((~same).sum())
    lpd = np.abs(da["log_post"][same] - db["log_post"][same])
    lpd = lpd[np.isfinite(lpd)]
    out = {"chains": chains, "steps": steps, "components": int(spec["P"]), "decisions": decisions, "chains_differing": differing,
           "first_flips": differing, "flips_per_1e9": (differing / decisions * 1e9) if decisions else None,
           "upper_95_per_1e9": ((3.0 if differing == 0 else differing + 2.0 * np.sqrt(differing) + 2.0) / decisions * 1e9) if decisions else None,
           "reference_geometry": {"lanes_per_chain": la["lanes_per_chain"], "block_threads": la["block_threads"]},
           "geometry": dict({"lanes_per_chain": lb["lanes_per_chain"], "block_threads": lb["block_threads"], "summation_order": lb.get("summation_order"), "kernel": lb.get("kernel")},
                            **{k: v for k, v in alt.items() if k != "lanes_per_chain"}),
           "lp_abs_diff_max": float(lpd.max()) if lpd.size else None, "lp_abs_diff_mean": float(lpd.mean()) if lpd.size else None,
           "lp_abs_typical": float(np.median(np.abs(da["log_post"][same]))) if same.any() else None,
           "expected_flips_bound": (2.0 * float(lpd.mean()) * decisions) if lpd.size else None,
           "differing_chain_ids": np.nonzero(~same)[0][:16].tolist()}
    return out


WORKLOADS = {
    # name: (family, n_obs, groups)
    "hier_n640_g8": ("hier_normal", 640, 8),
    "glm_n500": ("pois_glm", 500, 0),
    "cfg4_size": ("hier_normal", 10_000, 32),
    "normal_n1000": ("normal", 1_000, 0),
}


def user_spec_of(name):
    """a TRANSLATED closure of tests/js/user_models.js as a sampler spec (needs node): 'user:<closure>'"""
    import model_spec
    import user_host
    src, arrays, meta = user_host.translated(name)
    inf = float("inf")
    vec = lambda n: {"type": "real", "len": n, "top": n, "multidim": 1, "lower": -inf, "upper": inf}
    sca = lambda lo=-inf, hi=inf, ty="real": {"type": ty, "len": 1, "top": 1, "multidim": 0, "lower": lo, "upper": hi}
    if name in ("hier_normal_closure", "bench_hier"):
        G = 8 if name == "hier_normal_closure" else 32
        params, init = [vec(G), sca(), sca(0.0)], [0.5] * G + [0.5, 1.0]
    elif name in ("pois_glm_closure", "bench_glm"):
        n = 500 if name == "pois_glm_closure" else 50000
        params, init = [vec(8), sca(0.0, float(n - 1), "int")], [0.0] * 8 + [float(n // 2)]
    elif name == "bench_normal":
        params, init = [sca(), sca(0.0)], [0.5, 0.5]
    else:
        raise KeyError(name)
    return {"user": user_host.user_spec_part(src, arrays, meta), "params": params, "P": len(init), "init": init, "comp_opts": [dict(model_spec.DEFAULT_OPT) for _ in init]}


def spec_of(A, name, data_seed=20260925):
    import model_spec
    if name.startswith("user:"):
        return user_spec_of(name[5:])
    fam, n_obs, G = WORKLOADS[name]
    return model_spec.build_spec(fam, model_spec.make_data(fam, n_obs, data_seed, G=G or 32, exp=A.lib().amwg_exp))
