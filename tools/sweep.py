"""Geometry sweep on one GPU (development tool): param-updates/s of a config for several
lanes-per-chain / workgroup sizes.  Usage: python tools/sweep.py normal 10000 65536 [steps]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes.js_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import amwg_ctypes as A
import model_spec

model, n_obs, chains = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
combos = [tuple(map(int, c.split("x"))) for c in sys.argv[5:]] or [(0, 0), (1, 256), (1, 1024), (2, 1024), (4, 1024), (8, 1024), (16, 1024), (64, 1024), (4, 512)]
data = model_spec.make_data(model, n_obs, 20260925, G=32, exp=A.lib().amwg_exp)
spec = model_spec.build_spec(model, data)
for G, bt in combos:
    try:
        s = A.Sampler(spec, chains=chains, seed=1, lanes_per_chain=G, block_threads=bt)
        s.burn(3)
        t0 = time.time(); s.burn(steps); wall = time.time() - t0
        li = s.launch_info()
        ups = chains * steps * spec["P"] / (li["kernel_ms"] * 1e-3)
        print(f"{model} N={n_obs} C={chains} G={li['lanes_per_chain']} bt={li['block_threads']} grid={li['grid_blocks']} lds={li['lds_bytes']} "
              f"kernel_ms={li['kernel_ms']:.2f} wall_ms={wall*1e3:.2f} updates/s={ups:.3e}", flush=True)
        s.close()
    except Exception as e:
        print("FAIL", G, bt, e, flush=True)
