#!/usr/bin/env python3
"""tools/certified_campaign.py -- (GPU box) the certified decisions of round 5 (csrc/amwg_kernel.h) against the reference's expression,
counted: the same seeded job run by default (accept tests decided from a cheaper value of log_post with a rigorous bound: the Normal family at one lane per
chain, the Poisson family at 16 lanes, the hierarchical sweep kernel at 64) and with the expression, term by term IN THE REFERENCE'S ORDER, in every update
(options.full_evaluation = 1 at ONE lane per chain -- the order the certified kernels decide against at every lane count), chain against chain: final state,
accept / in-bounds counts, proposal scales, uniforms consumed and the cached log_post must agree in every bit -- a single wrongly certified decision would
change a chain for good.  Prints one JSON object; the
committed copy is profiles/r05_certified_campaign.json.

    python tools/certified_campaign.py [--scale 1.0] > gpurun_out/certified_campaign.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes.js_amd"), os.path.join(ROOT, "tests")]

import amwg_ctypes as A  # noqa: E402
import decision_parity as dp  # noqa: E402
import model_spec  # noqa: E402

# (workload: family, observations, groups), chains, steps, geometry
CAMPAIGN = [
    (("normal", 10_000, 0), 65_536, 20_000, {"lanes_per_chain": 1}),        # cfg2 itself: 2.6e9 decisions
    (("normal", 1_000, 0), 65_536, 40_000, {"lanes_per_chain": 1}),
    (("pois_glm", 50_000, 0), 8_192, 300, {"lanes_per_chain": 16}),         # cfg5 itself
    (("pois_glm", 500, 0), 16_384, 10_000, {"lanes_per_chain": 16}),
    (("hier_normal", 10_000, 32), 2_048, 4_000, {"lanes_per_chain": 64}),    # cfg4 itself (the one-lane run of 2 048 chains is 32 wavefronts: 14 ms per step)
    (("hier_normal", 640, 8), 16_384, 20_000, {"lanes_per_chain": 64}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0, help="multiplies the step counts")
    args = ap.parse_args()
    runs = []
    for (fam, n_obs, G), chains, steps, geom in CAMPAIGN:
        t0 = time.perf_counter()
        n = max(10, int(steps * args.scale))
        spec = model_spec.build_spec(fam, model_spec.make_data(fam, n_obs, 20260925, G=G or 32, exp=A.lib().amwg_exp))
        la, ia, da, sa = dp.run_one(A, spec, chains, n, 20260925, dict(geom, full_evaluation=0))
        t1 = time.perf_counter()
        lb, ib, db, sb = dp.run_one(A, spec, chains, n, 20260925, {"lanes_per_chain": 1, "full_evaluation": 1})
        t2 = time.perf_counter()
        same = np.all(ia["accepts"] == ib["accepts"], axis=0) & np.all(ia["inbounds"] == ib["inbounds"], axis=0)
        same &= da["uniforms"] == db["uniforms"]
        same &= np.all(sa.view(np.uint64) == sb.view(np.uint64), axis=0)
        same &= np.all(ia["prop_log_scale"].view(np.uint64) == ib["prop_log_scale"].view(np.uint64), axis=0)
        same &= da["log_post"].view(np.uint64) == db["log_post"].view(np.uint64)
        r = {"family": fam, "n_obs": n_obs, "chains": chains, "steps": n, "geometry": geom, "kernel": la.get("kernel"), "kernel_expression": lb.get("kernel"),
             "summation_order": la.get("summation_order"), "expression_lanes": lb.get("lanes_per_chain"),
             "decisions": int(ia["inbounds"].sum()), "chains_differing": int((~same).sum()), "seconds_default": t1 - t0, "seconds_expression": t2 - t1}
        runs.append(r)
        print("%-12s N=%-6d C=%-6d steps=%-6d %-24s decisions %.3g  differing %d  (%.1f s vs %.1f s)" %
              (fam, n_obs, chains, n, json.dumps(geom), r["decisions"], r["chains_differing"], r["seconds_default"], r["seconds_expression"]), file=sys.stderr)
    out = {"what": "same seed, same chain ids: default (certified decisions, at the listed lane count) vs options.full_evaluation = 1 at ONE lane per chain (the reference's expression in "
                   "the reference's order in every update); "
                   "a chain 'differs' when its final state, accept / in-bounds counts, proposal scales, uniform count or cached log_post differ in any bit",
           "version": A.lib().amwg_version().decode(), "runs": runs, "decisions_total": sum(r["decisions"] for r in runs),
           "chains_differing_total": sum(r["chains_differing"] for r in runs)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
