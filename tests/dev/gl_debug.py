#!/usr/bin/env python3
"""development aid (GPU box): group-local sampler vs the oracle's group-local mode and vs the one-lane run, several group counts; prints which
per-component counters differ."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (under tests/: it uses the oracle, which only the test tree may)
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes.js_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import amwg_ctypes as A, model_spec, oracle_lib

for (N, G) in ((640, 8), (1280, 16), (2000, 32), (10000, 32), (640, 64), (500, 5)):
    data = model_spec.make_data("hier_normal", N, 20260925, G=G)
    spec = model_spec.build_spec("hier_normal", data)
    a = A.Sampler(spec, chains=4, seed=5, lanes_per_chain=1)
    b = A.Sampler(spec, chains=4, seed=5, group_local=1)
    a.burn(120); b.burn(120)
    ia, ib = a.info(), b.info()
    o = oracle_lib.OracleChain(spec, 5, 0, lanes=64, group_local=True)
    o.burn(120)
    io = o.info()
    print("N %d G %d: state==1lane %s  state==oracleGL %s  accepts==1lane %s inbounds==1lane %s  accepts==oracle %s inbounds==oracle %s  uniforms %s/%s/%s lp %r %r" % (
        N, G, a.state()[:, 0].tobytes() == b.state()[:, 0].tobytes(), b.state()[:, 0].tobytes() == o.state().tobytes(),
        (ia["accepts"] == ib["accepts"]).all(), (ia["inbounds"] == ib["inbounds"]).all(),
        ib["accepts"][:, 0].tolist() == io["accepts"].tolist(), ib["inbounds"][:, 0].tolist() == io["inbounds"].tolist(),
        int(a.diag()["uniforms"][0]), int(b.diag()["uniforms"][0]), o.uniforms(), float(b.diag()["log_post"][0]), o.log_post()))
    if not (ia["accepts"] == ib["accepts"]).all() or not (ia["inbounds"] == ib["inbounds"]).all():
        print("   accepts 1lane", ia["accepts"][:, 0].tolist()); print("   accepts GL   ", ib["accepts"][:, 0].tolist())
        print("   inbound 1lane", ia["inbounds"][:, 0].tolist()); print("   inbound GL   ", ib["inbounds"][:, 0].tolist())
    a.close(); b.close()
