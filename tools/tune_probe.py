"""Development tool (GPU box): the lane count the static cost model picks vs the one AMWG_LANES_AUTOTUNE measures as fastest, for a few
families / sizes / chain counts -- where the two differ by more than the 12 % band the model needs another term (round 2: the hierarchical
family's lane-periodic group labels).   python tools/tune_probe.py"""
import sys; sys.path[:0]=["bayes.js_amd","tests"]
import amwg_ctypes as A, model_spec
for fam,n,ch in (("hier_normal",10000,16384),("hier_normal",10000,8192),("hier_normal",10000,4096),("pois_glm",50000,8192),("normal",10000,8192),("normal",10000,1024),("normal",1000,65536),("beta_bern",100000,16384)):
    spec=model_spec.build_spec(fam, model_spec.make_data(fam,n,20260925,G=32,exp=A.lib().amwg_exp))
    a=A.Sampler(spec, chains=ch, seed=1); la=a.launch_info()["lanes_per_chain"]; a.close()
    s=A.Sampler(spec, chains=ch, seed=1, lanes_per_chain=-2)
    t=dict(s.tuning()); best=min(t,key=t.get)
    print(fam, n, ch, "model picks", la, "tuned", s.launch_info()["lanes_per_chain"], "best", best, "model/best time ratio %.2f"%(t[la]/t[best]), {k:round(v,2) for k,v in t.items() if k<=64}, flush=True)
    s.close()
