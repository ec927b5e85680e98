#!/usr/bin/env python3
"""tools/reference_order_cost.py -- (GPU box) what one evaluation of the expression in the reference's order costs inside the certified multi-lane kernels
(csrc/amwg_models.h reference_order): the same job with the certified bounds widened 2^40-fold (options.test_bound_shift = 40: EVERY update is decided by that
evaluation) against the default, HIP-event time of the step kernels.  Also the guard's step: a hierarchical job whose chains start at
sigma = 1e-4 (|log_post| ~ 1e12: eta above 2^-7, every update falls back until sigma has grown).  Prints one JSON object (profiles/r05_reference_order_cost.json).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes.js_amd"), os.path.join(ROOT, "tests")]

import amwg_ctypes as A  # noqa: E402
import model_spec  # noqa: E402


def rate(spec, chains, lanes, steps, warm, **kw):
    s = A.Sampler(spec, chains=chains, seed=20260925, lanes_per_chain=lanes, steps_per_launch=steps, **kw)
    s.burn(warm)
    s.burn(steps)
    li = s.launch_info()
    s.close()
    return {"kernel": li["kernel"], "kernel_ms": li["kernel_ms"], "steps": steps, "chains": chains, "us_per_update_per_chain": li["kernel_ms"] * 1e3 / (steps * spec["P"])}


def main():
    out = {"version": A.lib().amwg_version().decode(), "what": "us_per_update_per_chain = kernel time of a launch / (steps x components): the latency of one update of one chain "
           "(all chains run side by side, at most two wavefronts per SIMD).  shift40 - default = the cost of deciding an update by reference_order (1 .. 2 evaluations)"}
    hs = model_spec.build_spec("hier_normal", model_spec.make_data("hier_normal", 10_000, 20260925, G=32))
    out["cfg4"] = {"default": rate(hs, 2048, 64, 100, 300), "shift40": rate(hs, 2048, 64, 10, 10, test_bound_shift=40)}
    gs = model_spec.build_spec("pois_glm", model_spec.make_data("pois_glm", 50_000, 20260925, exp=A.lib().amwg_exp))
    out["cfg5"] = {"default": rate(gs, 8192, 16, 20, 40), "shift40": rate(gs, 8192, 16, 2, 2, test_bound_shift=40)}
    # the guard's step: every chain starts at sigma = 1e-4
    s = A.Sampler(hs, chains=2048, seed=20260925, lanes_per_chain=64, steps_per_launch=25)
    st = s.state()
    st[-1] = 1e-4
    s.set_state(st)
    phases = []
    for k in range(12):
        s.burn(25)
        phases.append({"steps": (k + 1) * 25, "kernel_ms": s.launch_info()["kernel_ms"], "median_sigma": float(np.median(s.state()[-1])),
                       "max_abs_log_post": float(np.max(np.abs(s.diag()["log_post"])))})
    s.close()
    out["guard_step"] = {"what": "cfg4, 2048 chains, all started at sigma = 1e-4; launches of 25 steps (an adapted launch of 25 steps takes ~0.42 ms)", "launches": phases}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
