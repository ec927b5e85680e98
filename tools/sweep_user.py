"""Geometry sweep of a TRANSLATED closure on one GPU (development tool), next to the built-in family where one exists.
Usage: python tools/sweep_user.py bench_normal 65536 [steps] [GxBT ...]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes.js_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import amwg_ctypes as A
import user_host

name, chains = sys.argv[1], int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
combos = [tuple(map(int, c.split("x"))) for c in sys.argv[4:]] or [(0, 0)]
src, arrays, meta = user_host.translated(name)
if os.environ.get("SWEEP_UNFUSE"):      # A/B: the softplus of a logistic likelihood as two calls (what the translator emitted before log1p_exp_v8; translate with
    import re                           # AMWG_TRANSLATE_OPTS='{"no_open_softplus":true}' so that the call is the plain one)
    src = re.sub(r"log1p_exp_v8\((v_\w+)\)", r"log1p_v8(exp_v8(\1))", src)
inf = float("inf")
LAYOUT = {  # completed params of the bench closures
    "bench_normal": ([("real", 1, -inf, inf, 0.5), ("real", 1, 0.0, inf, 0.5)]),
    "bench_normal_50k": ([("real", 1, -inf, inf, 0.5), ("real", 1, 0.0, inf, 0.5)]),
    "bench_normal_n65": ([("real", 1, -inf, inf, 0.5), ("real", 1, 0.0, inf, 0.5)]),
    "bench_normal_expr": ([("real", 1, -inf, inf, 0.5), ("real", 1, -inf, inf, 0.5), ("real", 1, 0.0, inf, 1.0)]),
    "bench_bern": ([("real", 1, 0.0, 1.0, 0.5)]),
    "bench_hier": ([("real", 32, -inf, inf, 0.5), ("real", 1, -inf, inf, 0.5), ("real", 1, 0.0, inf, 1.0)]),
    "bench_glm": ([("real", 8, -inf, inf, 0.0), ("int", 1, 0.0, 49999.0, 25000.0)]),
    "pois_const_rate": ([("real", 1, 0.0, inf, 2.0), ("real", 1, -inf, inf, 0.0)]),      # N = 1e5 counts, constant rate: K-valued fast-forward with one lane per chain
    "binom_const_size": ([("real", 1, 0.0, 1.0, 0.5)]),
}
if name in LAYOUT:
    params, init = [], []
    for ty, ln, lo, hi, iv in LAYOUT[name]:
        params.append({"type": ty, "len": ln, "top": ln, "multidim": 0 if ln == 1 else 1, "lower": lo, "upper": hi})
        init += [iv] * ln
else:      # any fixture closure with a golden: the completed parameters the reference built
    import golden_io
    params, init = [], []
    for p in golden_io.load("user_" + name)["chains"][0]["params_completed"]:
        ln = int(np.prod(p["dim"]))
        params.append({"type": p["type"], "len": ln, "top": p["dim"][0], "multidim": 0 if p["dim"] == [1] else 1, "lower": p["lower"], "upper": p["upper"]})
        init += p["init"]
opt = {"prop_log_scale": 0.0, "batch_size": 50, "max_adaptation": 0.33, "initial_adaptation": 1.0, "target_accept_rate": 0.44, "is_adapting": True}
spec = {"user": user_host.user_spec_part(src, arrays, meta),
        "params": params, "P": len(init), "init": init, "comp_opts": [dict(opt) for _ in init]}
for G, bt in combos:
    try:
        t0 = time.time()
        s = A.Sampler(spec, chains=chains, seed=1, lanes_per_chain=G, block_threads=bt)
        tc = time.time() - t0
        s.burn(3)
        s.burn(steps)
        li = s.launch_info()
        ups = chains * steps * spec["P"] / (li["kernel_ms"] * 1e-3)
        print(f"{name} (translated) C={chains} G={li['lanes_per_chain']} bt={li['block_threads']} grid={li['grid_blocks']} lds={li['lds_bytes']} "
              f"create_s={tc:.2f} kernel_ms={li['kernel_ms']:.2f} updates/s={ups:.3e}", flush=True)
        s.close()
    except Exception as e:
        print("FAIL", G, bt, e, flush=True)
