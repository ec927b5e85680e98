#!/usr/bin/env python3
"""tools/bound_audit.py -- are the bounds of the certified decisions RIGOROUS, not merely never-yet-wrong?  (needs a GPU)

The certified kernels (csrc/amwg_kernel.h kCert: amwg_step_kernel_cert<NormalModel,1,..>, <PoisGlmModel,16,..>, amwg_sweep_kernel_cert<HierNormalModel,..>)
decide the accept test exp(prop - curr) > u (mcmc.js:527-528) from a cheaper value A of log_post and a hand-derived bound eps on |A - E|, E = the reference's
expression.  The round-5 campaigns compared final states over 1e10 decisions; they cannot see a bound that is too small by a factor, because the actual
|A - E| is orders of magnitude below eps and a wrong verdict then needs a uniform inside a ~1e-12-wide window.  This tool runs the AUDIT build of the library
(csrc/libamwg_audit.so, -DAMWG_AUDIT: the same kernels evaluating E beside A in EVERY update, include/amwg_selftest.h amwg_audit_fetch) and reports, per case,

    max |A - E| / eps        over every audited value          (a bound holds iff <= 1; the review asks for <= 0.5)
    max |dA - dE| / eta      over every audited difference     (dE = RN(E_prop - E_cur), what the reference's test takes the exponential of)
    wrong verdicts           certified verdicts that contradict exp_v8(dE) > u     (must be 0)
    histograms of both ratios by binary exponent

on the BASELINE configurations at full size and on adversarial inputs: data far from the origin (x = 1e8 + noise: n c and Q cancel), sigma at 1e-6 and 1e6, n in
{1, 2, 17, 63, 65}, hierarchical rows with one or no observation, a group count that is not a power of two (update-by-update path), Poisson predictors next to the
690 cut-off and counts up to ~1e6.  `--shrink` adds the converse experiment: the bounds multiplied by 2^-k (test_bound_shift < 0, audit build only) until a
verdict goes wrong or a chain differs from the expression-in-every-update run.

    python tools/bound_audit.py --out profiles/r06_bound_audit.json            # everything (a few minutes of GPU)
    python tools/bound_audit.py --quick                                        # the small cases only (what tests/test_gpu_bound_audit.py asserts)
"""
import argparse
import ctypes as C
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AUDIT_LIB = os.path.join(ROOT, "bayes.js_amd", "csrc", "libamwg_audit.so")
os.environ["AMWG_LIB"] = AUDIT_LIB      # (before the binding is imported: it reads the variable once)
for p in (ROOT, os.path.join(ROOT, "bayes.js_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402

import amwg_ctypes as A  # noqa: E402
import model_spec  # noqa: E402

INF = float("inf")


def audit_fetch(s, reset=False):
    L = A.lib()
    L.amwg_audit_fetch.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int32]
    per = np.zeros((4, s.C))
    hist = np.zeros(128, dtype=np.uint64)
    rc = L.amwg_audit_fetch(s.h, per.ctypes.data_as(C.POINTER(C.c_double)), hist.ctypes.data_as(C.POINTER(C.c_uint64)), int(reset))
    if rc != 0:
        raise RuntimeError(L.amwg_last_error().decode())
    return per, hist.reshape(2, 64)


def hist_summary(h):
    """{'2^k': count} for the occupied bins; bin b >= 1 holds ratios in [2^(b-40), 2^(b-39)), bin 0 = exactly equal, bin 63 also NaN"""
    out = {}
    for b in range(64):
        if h[b]:
            out["equal" if b == 0 else "2^%d" % (b - 40)] = int(h[b])
    return out


def normal_case(name, x, chains, steps, state=None, hyper=None, seed=11, suff=0):
    spec = model_spec.build_spec("normal", {"x": np.asarray(x, dtype=np.float64)}, hyper=hyper)
    return dict(name=name, spec=spec, chains=chains, steps=steps, lanes=1, state=state, seed=seed, suff=suff)


def hier_case(name, y, g, G, chains, steps, state=None, hyper=None, seed=12):
    spec = model_spec.build_spec("hier_normal", {"x": np.asarray(y, dtype=np.float64), "g": np.asarray(g, dtype=np.int32), "G": G}, G=G, hyper=hyper)
    return dict(name=name, spec=spec, chains=chains, steps=steps, lanes=64, state=state, seed=seed)


def pois_case(name, X, y, chains, steps, state=None, seed=13):
    spec = model_spec.build_spec("pois_glm", {"x": np.asarray(X, dtype=np.float64).reshape(-1), "y": np.asarray(y, dtype=np.float64), "K": 7})
    return dict(name=name, spec=spec, chains=chains, steps=steps, lanes=16, state=state, seed=seed)


def glm_data(n, rng, intercept, scale=0.5, cp_frac=0.4, shift=0.3):
    X = np.empty((n, 7))
    X[:, 0] = 1.0
    X[:, 1:] = scale * rng.standard_normal((n, 6))
    beta = np.array([intercept, 0.2, -0.1, 0.05, 0.1, -0.2, 0.15])
    eta = X @ beta + np.where(np.arange(n) >= int(cp_frac * n), shift, 0.0)
    return X, rng.poisson(np.exp(eta)).astype(np.float64)


def cases(quick):
    rng = np.random.default_rng(20260925)
    out = []
    # ---- Normal family, one lane per chain (csrc/amwg_models.h NormalModel::log_post_approx)
    small = 256 if quick else 4096
    x1k = model_spec.make_data("normal", 1000, 20260925)["x"]
    out.append(normal_case("normal_n1000", x1k, small, 300))
    for n in (1, 2, 17, 63, 65):
        out.append(normal_case("normal_n%d" % n, 3.0 + 2.0 * rng.standard_normal(n), small, 300))
    xfar = 1e8 + rng.standard_normal(1000)
    out.append(normal_case("normal_x1e8_start_far", xfar, small, 300))                                          # mu starts at 0.5: Q ~ 1e19 swamps n c
    out.append(normal_case("normal_x1e8_start_near", xfar, small, 300, state=[1e8, 1.0]))                      # mu next to the data: x - mu exact, n c and Q comparable
    out.append(normal_case("normal_sigma_1e-6", x1k, small, 200, state=[3.0, 1e-6]))
    out.append(normal_case("normal_sigma_1e6", x1k, small, 200, state=[3.0, 1e6]))
    out.append(normal_case("normal_tiny_data", 1e-250 * (1.0 + rng.random(1000)), small, 200))
    out.append(normal_case("normal_tight_prior", x1k, small, 300, hyper=[0, 1e-3, 0, 100]))                    # |prior| dominates mag
    out.append(normal_case("normal_constant_data", np.full(513, 7.25), small, 300))                             # S2 = 0 at mu = 7.25
    # ... and the opt-in third tier (amwg_options::sufficient_statistics: S2 = SS + n (xbar - mu)^2, no pass): the same bound on the same inputs
    out.append(normal_case("suffstat_n1000", x1k, small, 300, suff=1))
    out.append(normal_case("suffstat_n1", 3.0 + 2.0 * rng.standard_normal(1), small, 300, suff=1))
    out.append(normal_case("suffstat_n65", 3.0 + 2.0 * rng.standard_normal(65), small, 300, suff=1))
    out.append(normal_case("suffstat_x1e8_start_far", xfar, small, 300, suff=1))
    out.append(normal_case("suffstat_x1e8_start_near", xfar, small, 300, state=[1e8, 1.0], suff=1))           # xbar - mu: the double-double matters here
    out.append(normal_case("suffstat_sigma_1e-6", x1k, small, 200, state=[3.0, 1e-6], suff=1))
    out.append(normal_case("suffstat_constant_data", np.full(513, 7.25), small, 300, suff=1))                   # SS = 0
    out.append(normal_case("suffstat_tiny_data", 1e-250 * (1.0 + rng.random(1000)), small, 200, suff=1))
    # ---- hierarchical family, the sweep kernel (HierNormalModel::sweep_approx / log_post_approx / value_bound / difference_bound)
    hs = 64 if quick else 512
    d = model_spec.make_data("hier_normal", 640, 20260925, G=8)
    out.append(hier_case("hier_n640_g8", d["x"], d["g"], 8, hs, 120))
    # (one observation per lane; a ragged last round; lanes without any; 2 to 64 groups.  A group count that is not a power of two has no row layout -- its labels do not
    # repeat with the lane stride -- and runs the plain kernel: nothing certified, nothing to audit)
    for n, G in ((64, 32), (65, 32), (100, 4), (128, 2), (1024, 16), (64, 64)):
        dd = model_spec.make_data("hier_normal", n, 20260925 + n, G=G)
        out.append(hier_case("hier_n%d_g%d" % (n, G), dd["x"], dd["g"], G, hs, 80))
    g8 = np.arange(640) % 8
    out.append(hier_case("hier_y1e8", 1e8 + rng.standard_normal(640) + 3.0 * rng.standard_normal(8)[g8], g8, 8, hs, 120))
    out.append(hier_case("hier_sigma_1e-4", d["x"], d["g"], 8, hs, 80, state=[0.5] * 8 + [0.5, 1e-4]))
    out.append(hier_case("hier_sigma_1e4", d["x"], d["g"], 8, hs, 80, state=[0.5] * 8 + [0.5, 1e4]))
    out.append(hier_case("hier_pinned_sigma", d["x"], d["g"], 8, hs, 120, hyper=[0, 100, 0, 1, 10]))            # sigma ~ unif(0, 1): pressed against its upper bound
    # ---- Poisson family, 16 lanes per chain (PoisGlmModel::log_post_approx)
    ps = 64 if quick else 256
    dg = model_spec.make_data("pois_glm", 500, 20260925)
    Xs = dg["x"].reshape(500, 7)
    out.append(pois_case("pois_n500", Xs, dg["y"], ps, 150))
    for n in (1, 2, 17, 63, 65):
        Xn, yn = glm_data(n, rng, 0.5)
        out.append(pois_case("pois_n%d" % n, Xn, yn, ps, 150))
    Xb, yb = glm_data(400, rng, 13.0, scale=0.1)                                                                # counts ~ 4e5 .. 1e6
    out.append(pois_case("pois_counts_1e6", Xb, yb, ps, 150, state=[13.0, 0.2, -0.1, 0.05, 0.1, -0.2, 0.15, 0.3, 160.0]))
    Xo = np.zeros((300, 7)); Xo[:, 0] = 1.0; Xo[:, 1:] = 1e-3 * rng.standard_normal((300, 6))
    out.append(pois_case("pois_H_689", Xo, rng.poisson(50.0, 300).astype(np.float64), ps, 100, state=[689.2] + [0.0] * 7 + [100.0]))      # H within 1 of the 690 cut-off: the bound is finite ...
    out.append(pois_case("pois_H_691", Xo, rng.poisson(50.0, 300).astype(np.float64), ps, 100, state=[690.6] + [0.0] * 7 + [100.0]))      # ... and beyond it the expression decides (nothing audited while H > 690)
    out.append(pois_case("pois_zero_counts", Xs, np.zeros(500), ps, 120))
    # ---- a TRANSLATED closure with a certified tail (translate.js tailPlan, csrc/amwg_user.h norm_tail_approx): BASELINE configs[1] as a plain closure; hiprtc compiles
    # its kernel with -DAMWG_AUDIT in the audit build
    if shutil.which("node"):
        import user_host
        src, arrays, meta = user_host.translated("bench_normal")
        inf = float("inf")
        uparams = [{"type": "real", "len": 1, "top": 1, "multidim": 0, "lower": -inf, "upper": inf}, {"type": "real", "len": 1, "top": 1, "multidim": 0, "lower": 0.0, "upper": inf}]
        uspec = {"user": user_host.user_spec_part(src, arrays, meta), "params": uparams, "P": 2, "init": [0.5, 0.5], "comp_opts": [dict(model_spec.DEFAULT_OPT) for _ in range(2)], "n_obs": 10000}
        out.append(dict(name="user_bench_normal", spec=uspec, chains=512 if quick else 65536, steps=150, lanes=1, state=None, seed=14))
        out.append(dict(name="user_bench_normal_sigma_1e-5", spec=uspec, chains=512, steps=100, lanes=1, state=[3.0, 1e-5], seed=15))
        # ... and one with a ROW PLAN the translator marked kRowCert (csrc/amwg_rows.h: amwg_user_sweep_cert): BASELINE configs[3] as a plain closure, and a small one
        for nm, nobs, G, ch, st in (("hier_normal_closure", 640, 8, 64 if quick else 512, 120), ("bench_hier", 10000, 32, 64 if quick else 2048, 40 if quick else 60)):
            hsrc, harrays, hmeta = user_host.translated(nm) if nm == "bench_hier" else (user_host.host_model(nm).source, user_host.host_model(nm).arrays, user_host.host_model(nm).meta)
            hparams = [{"type": "real", "len": G, "top": G, "multidim": 1, "lower": -inf, "upper": inf}, {"type": "real", "len": 1, "top": 1, "multidim": 0, "lower": -inf, "upper": inf},
                       {"type": "real", "len": 1, "top": 1, "multidim": 0, "lower": 0.0, "upper": inf}]
            hspec = {"user": user_host.user_spec_part(hsrc, harrays, hmeta), "params": hparams, "P": G + 2, "init": [0.5] * G + [0.5, 1.0],
                     "comp_opts": [dict(model_spec.DEFAULT_OPT) for _ in range(G + 2)], "n_obs": nobs}
            out.append(dict(name="user_" + nm, spec=hspec, chains=ch, steps=st, lanes=64, state=None, seed=16))
        # ... and one with a certified POISSON TAIL (translate.js poisTailPlan, csrc/amwg_ptail.h pois_tail_approx: amwg_user_step_cert at 16 lanes per chain): the small
        # closure of the goldens, the same pressed against the 690 cut-off's neighbourhood by a large intercept, and BASELINE configs[4] as a plain closure
        for nm, nobs, ch, st, state in (("pois_glm_closure", 500, 64 if quick else 256, 150, None), ("pois_glm_closure", 500, 64, 100, [3.0, 0.2, -0.1, 0.05, 0.1, -0.2, 0.15, 0.3, 160.0]),
                                        ("bench_glm", 50000, 64 if quick else 2048, 20 if quick else 40, None)):
            psrc, parrays, pmeta = user_host.translated(nm)
            pparams = [{"type": "real", "len": 8, "top": 8, "multidim": 1, "lower": -inf, "upper": inf}, {"type": "int", "len": 1, "top": 1, "multidim": 0, "lower": 0.0, "upper": float(nobs - 1)}]
            pspec = {"user": user_host.user_spec_part(psrc, parrays, pmeta), "params": pparams, "P": 9, "init": [0.0] * 8 + [float(nobs // 2)],
                     "comp_opts": [dict(model_spec.DEFAULT_OPT) for _ in range(9)], "n_obs": nobs}
            out.append(dict(name="user_" + nm + ("_large_rates" if state else ""), spec=pspec, chains=ch, steps=st, lanes=16, state=state, seed=17))
        # ... and the tail's fallback paths (amwg_ptail.h: eta by the closure's own statements with H taken over the rows; per-lane state reads; no row cache)
        for nm in ("pois_tail_nonlinear", "pois_tail_gather", "pois_tail_next_row"):
            psrc, parrays, pmeta = user_host.translated(nm)
            pparams = [{"type": "real", "len": 8, "top": 8, "multidim": 1, "lower": -inf, "upper": inf}, {"type": "int", "len": 1, "top": 1, "multidim": 0, "lower": 0.0, "upper": 516.0}]
            pspec = {"user": user_host.user_spec_part(psrc, parrays, pmeta), "params": pparams, "P": 9, "init": [0.1] * 8 + [250.0], "comp_opts": [dict(model_spec.DEFAULT_OPT) for _ in range(9)], "n_obs": 517}
            out.append(dict(name="user_" + nm, spec=pspec, chains=64 if quick else 256, steps=120, lanes=16, state=None, seed=18))
    if not quick:
        # ---- BASELINE configs at full size
        out.append(normal_case("cfg2_full", model_spec.make_data("normal", 10000, 20260925)["x"], 65536, 150, seed=20260925))
        out.append(normal_case("cfg2_full_suffstat", model_spec.make_data("normal", 10000, 20260925)["x"], 65536, 150, seed=20260925, suff=1))
        d4 = model_spec.make_data("hier_normal", 10000, 20260925, G=32)
        out.append(hier_case("cfg4_full", d4["x"], d4["g"], 32, 2048, 60, seed=20260925))
        d5 = model_spec.make_data("pois_glm", 50000, 20260925)
        out.append(pois_case("cfg5_full", d5["x"].reshape(50000, 7), d5["y"], 8192, 40, seed=20260925))
    return out


def run_case(c, shift=0, full_evaluation=0):
    s = A.Sampler(c["spec"], chains=c["chains"], seed=c["seed"], lanes_per_chain=c["lanes"], test_bound_shift=shift, full_evaluation=full_evaluation,
                  sufficient_statistics=(c.get("suff", 0) if full_evaluation == 0 else 0))
    try:
        kernel = s.launch_info()["kernel"]
        if c["state"] is not None:
            st = np.repeat(np.asarray(c["state"], dtype=np.float64)[:, None], c["chains"], axis=1)
            s.set_state(st)
        t0 = time.time()
        s.burn(c["steps"])
        per, hist = (audit_fetch(s) if full_evaluation == 0 else (np.zeros((4, c["chains"])), np.zeros((2, 64), dtype=np.uint64)))
        wall = time.time() - t0
        info = s.info()
        final = (s.state().tobytes(), info["accepts"].tobytes(), s.diag()["uniforms"].tobytes())
    finally:
        s.close()
    return dict(kernel=kernel, per=per, hist=hist, wall=wall, final=final, accepts=int(info["accepts"].sum()), inbounds=int(info["inbounds"].sum()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--shrink", action="store_true", help="also: bounds x 2^-k until a verdict goes wrong / a chain differs (three small cases)")
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    if not os.path.exists(AUDIT_LIB):
        raise SystemExit("libamwg_audit.so is missing: make -C bayes.js_amd/csrc libamwg_audit.so")
    rec = {"library": A.lib().amwg_version().decode(), "cases": [], "shrink": []}
    worst_v = worst_d = 0.0
    wrong = 0
    for c in cases(a.quick):
        if a.only and a.only not in c["name"]:
            continue
        r = run_case(c)
        per = r["per"]
        mv, md = float(np.nanmax(per[0])) if not np.isnan(per[0]).all() else float("nan"), float(np.nanmax(per[1])) if not np.isnan(per[1]).all() else float("nan")
        if np.isnan(per[0]).any() or np.isnan(per[1]).any():
            mv = md = float("nan")
        e = dict(name=c["name"], kernel=r["kernel"], chains=c["chains"], steps=c["steps"], n_obs=c["spec"]["n_obs"], in_bounds_proposals=r["inbounds"],
                 audited_decisions=int(per[2].sum()), wrong_verdicts=int(per[3].sum()), max_value_ratio=mv, max_difference_ratio=md,
                 value_ratio_hist=hist_summary(r["hist"][0]), difference_ratio_hist=hist_summary(r["hist"][1]), seconds=round(r["wall"], 2))
        rec["cases"].append(e)
        if "_cert" in r["kernel"]:
            worst_v, worst_d, wrong = max(worst_v, mv if mv == mv else INF), max(worst_d, md if md == md else INF), wrong + e["wrong_verdicts"]
        print("%-26s %-48s audited %11d  wrong %d  max |A-E|/eps %.3g  max |dA-dE|/eta %.3g  (%.1f s)" % (c["name"], r["kernel"], e["audited_decisions"], e["wrong_verdicts"], mv, md, r["wall"]), flush=True)
    if a.shrink:
        # The converse experiment.  (a) natural uniforms, bounds x 2^-k: when does a verdict go wrong / a chain leave the expression-in-every-update run?  (Expected: never --
        # a wrong verdict needs a uniform within |dA - dE| ~ 1e-12 of exp(dA); the point of recording it.)  (b) ADVERSARIAL uniforms (AMWG_AUDIT_ADVERSARIAL=1: every certified
        # decision's uniform replaced by one 1.5 eta off exp(dA), alternately on either side -- the sliver's edge, where a bound too small by more than 1.5 MUST produce a
        # wrong verdict): the first k with wrong verdicts measures how much room the bound really has, and shows that the audit's detector fires at all.
        by = {c["name"]: c for c in cases(True)}
        for name in ("normal_n1000", "normal_constant_data", "hier_n640_g8", "pois_n500", "pois_zero_counts") + (("user_pois_glm_closure",) if "user_pois_glm_closure" in by else ()):
            c = dict(by[name])
            c["chains"], c["steps"] = (4096, 400) if name.startswith("normal") else (512, 150)
            ref = run_case(c, full_evaluation=1)      # the expression in every update (the multi-lane families: compared through accept counts and uniforms consumed)
            row = {"name": name, "decisions_per_run": ref["inbounds"], "natural_uniforms": [], "adversarial_uniforms": [], "first_shift_with_wrong_verdicts_adversarial": None}
            for k in range(0, -61, -6):
                r = run_case(c, shift=k)
                same = r["final"][1:] == ref["final"][1:] and (r["final"][0] == ref["final"][0] or not name.startswith("normal"))
                row["natural_uniforms"].append({"shift": k, "wrong_verdicts": int(r["per"][3].sum()), "decisions_equal_expression_run": bool(same)})
            os.environ["AMWG_AUDIT_ADVERSARIAL"] = "1"
            try:
                for k in list(range(0, -13, -1)) + list(range(-16, -41, -4)):
                    r = run_case(c, shift=k)
                    w = int(r["per"][3].sum())
                    row["adversarial_uniforms"].append({"shift": k, "audited": int(r["per"][2].sum()), "wrong_verdicts": w})
                    if w and row["first_shift_with_wrong_verdicts_adversarial"] is None:
                        row["first_shift_with_wrong_verdicts_adversarial"] = k
            finally:
                os.environ["AMWG_AUDIT_ADVERSARIAL"] = "0"
            print("shrink %-22s natural uniforms: wrong verdicts %s, chains leaving the expression run: %s;  adversarial uniforms: first wrong verdicts at 2^%s (%s)" % (
                name, sum(q["wrong_verdicts"] for q in row["natural_uniforms"]), sum(not q["decisions_equal_expression_run"] for q in row["natural_uniforms"]),
                row["first_shift_with_wrong_verdicts_adversarial"], [(q["shift"], q["wrong_verdicts"]) for q in row["adversarial_uniforms"]][:14]), flush=True)
            rec["shrink"].append(row)
    rec["summary"] = {"max_value_ratio": worst_v, "max_difference_ratio": worst_d, "wrong_verdicts": wrong,
                      "bounds_hold_with_factor_two": bool(worst_v <= 0.5 and worst_d <= 0.5 and wrong == 0)}
    print(json.dumps(rec["summary"]))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(rec, f, indent=1)
    return 0 if rec["summary"]["bounds_hold_with_factor_two"] else 1


if __name__ == "__main__":
    sys.exit(main())
