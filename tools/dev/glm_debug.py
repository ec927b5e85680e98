#!/usr/bin/env python3
"""development aid (GPU box): per-chain posterior means of the N = 500 Poisson GLM at 64 lanes, chains 1000.. under seed + 1 (the two-sample moment test)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes.js_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import amwg_ctypes as A, model_spec
data = model_spec.make_data("pois_glm", 500, 20260925, exp=A.lib().amwg_exp)
spec = model_spec.build_spec("pois_glm", data)
for chains in (16, 2048):
    s = A.Sampler(spec, chains=chains, seed=20260926, chain_offset=1000, lanes_per_chain=64)
    s.burn(5000)
    d = s.sample(15000, 5)
    m7 = d[:, 7, :].mean(axis=0); mcp = d[:, 8, :].mean(axis=0); scp = d[:, 8, :].std(axis=0)
    print("chains", chains, "pooled beta7 %.4f  cp %.1f; library moments:" % (d[:, 7, :].mean(), d[:, 8, :].mean()), s.moments()[0][7:9])
    print("  first 16 per-chain beta7:", m7[:16].round(4).tolist())
    print("  first 16 per-chain cp mean:", mcp[:16].round(1).tolist())
    print("  quantiles of per-chain beta7 mean:", np.quantile(m7, [0, .05, .25, .5, .75, .95, 1]).round(4).tolist())
    print("  share of chains with beta7 mean < 0.15:", float((m7 < 0.15).mean()), " cp sd > 80:", float((scp > 80).mean()))
    s.close()
