#!/usr/bin/env python3
"""tools/flip_rate.py -- the decision-parity campaign (GPU box): the same seeded job with one lane per chain (the reference's summation order)
and at 64 / 32 / 16 lanes / group-local, counting the chains whose run ever differs (tests/decision_parity.py).  Two groups: the kernels that decide
against the expression in the reference's own order (amwg_summation_order() == 1: the defaults of cfg4 and cfg5 since round 5) must not differ in ANY chain,
cached log_post included; the others (their own lane order) get a rate.  Prints one JSON object; the committed copy is profiles/r05_flip_rate.json, quoted in
DESIGN.md section 2 and in the bench line's parity.flip_rate.

    python tools/flip_rate.py [--scale 1.0] > gpurun_out/flip_rate.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes.js_amd"), os.path.join(ROOT, "tests")]

import amwg_ctypes as A  # noqa: E402
import decision_parity as dp  # noqa: E402

CAMPAIGN = [
    ("hier_n640_g8", 65_536, 20_000, {"lanes_per_chain": 64}),
    ("hier_n640_g8", 65_536, 20_000, {"lanes_per_chain": 64, "group_local": 1}),
    ("glm_n500", 16_384, 10_000, {"lanes_per_chain": 64}),
    ("glm_n500", 16_384, 10_000, {"lanes_per_chain": 16}),      # (round 5: the default geometry of cfg5 -- certified decisions, four chains to a wavefront)
    ("cfg4_size", 16_384, 4_000, {"lanes_per_chain": 64}),
    ("cfg4_size", 16_384, 4_000, {"lanes_per_chain": 64, "group_local": 1}),
    ("cfg4_size", 16_384, 4_000, {"lanes_per_chain": 32}),
    ("normal_n1000", 65_536, 20_000, {"lanes_per_chain": 64}),
    # (round 6) TRANSLATED closures whose certified kernels decide in the reference's order (translate.js rowPlan / poisTailPlan; docs/CERTIFIED.md): against the same closure
    # at one lane per chain -- not one chain may differ, cached log_post included
    ("user:hier_normal_closure", 65_536, 10_000, {"lanes_per_chain": 64}),
    ("user:pois_glm_closure", 16_384, 10_000, {"lanes_per_chain": 16}),
    ("user:bench_hier", 4_096, 2_000, {"lanes_per_chain": 64}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0, help="multiplies the step counts")
    args = ap.parse_args()
    runs = []
    ref_runs = {}      # the one-lane reference run of a (workload, chains, steps) is made once
    import shutil
    for wl, chains, steps, alt in CAMPAIGN:
        if wl.startswith("user:") and shutil.which("node") is None:
            continue
        t0 = time.perf_counter()
        n = max(10, int(steps * args.scale))
        spec = dp.spec_of(A, wl)
        if (wl, chains, n) not in ref_runs:
            ref_runs = {(wl, chains, n): dp.run_one(A, spec, chains, n, 20260925, {"lanes_per_chain": 1})}
        r = dp.compare(A, spec, chains, n, seed=20260925, alt=alt, ref_run=ref_runs[(wl, chains, n)])
        r["workload"] = wl
        r["seconds"] = time.perf_counter() - t0
        runs.append(r)
        print("%-14s %-44s decisions %.3g  differing %d  lp diff max %.3g mean %.3g  (%.1f s)" %
              (wl, json.dumps(alt), r["decisions"], r["chains_differing"], r["lp_abs_diff_max"] or -1, r["lp_abs_diff_mean"] or -1, r["seconds"]), file=sys.stderr)
    ref = [r for r in runs if r["geometry"].get("summation_order") == 1]
    lane = [r for r in runs if r["geometry"].get("summation_order") != 1]
    tot_d = sum(r["decisions"] for r in lane)
    tot_f = sum(r["first_flips"] for r in lane)
    out = {"what": "same seed, same chain ids: one lane per chain (reference order) vs the listed geometry; a chain 'differs' when its final state, accept / "
                   "in-bounds counts, proposal scales or uniform count differ (mcmc.js:527-528: the accept test is the only place the summation order can matter).  "
                   "reference_order: the kernels that decide against the expression in the reference's order (summation_order 1) -- not one chain may differ, cached "
                   "log_post included; decisions_total / flips_per_1e9: the kernels that sum in their own lane order",
           "version": A.lib().amwg_version().decode(), "runs": runs,
           "reference_order": {"decisions_total": sum(r["decisions"] for r in ref), "chains_differing": sum(r["chains_differing"] for r in ref),
                               "log_post_differs": any((r["lp_abs_diff_max"] or 0.0) != 0.0 for r in ref),
                               "geometries": [dict(r["geometry"], workload=r["workload"]) for r in ref]},
           "decisions_total": tot_d, "first_flips_total": tot_f,
           "flips_per_1e9": tot_f / tot_d * 1e9, "upper_95_per_1e9": (3.0 if tot_f == 0 else tot_f + 2.0 * tot_f ** 0.5 + 2.0) / tot_d * 1e9}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
