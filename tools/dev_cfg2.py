import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes.js_amd"), os.path.join(ROOT, "tests")]
import numpy as np, amwg_ctypes as A, model_spec
d = model_spec.make_data("normal", 10000, 20260925)
spec = model_spec.build_spec("normal", d)
SUFF = int(os.environ.get("SUFF", "0"))
s = A.Sampler(spec, chains=int(os.environ.get("CHAINS", "65536")), seed=1, steps_per_launch=100, sufficient_statistics=SUFF)
s.burn(1000)
best = 1e9
for rep in range(8):
    s.burn(100); best = min(best, s.launch_info()["kernel_ms"])
print(os.environ.get("AMWG_LIB", "product"), s.launch_info()["kernel"], "best %.3f ms per 100 steps -> %.4g updates/s" % (best, s.C * 200 / best * 1e3), flush=True)
