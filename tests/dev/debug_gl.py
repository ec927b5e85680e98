"""Development aid (GPU box): step the group-local GPU sampler and its oracle side by side, print the first step at which they part."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (under tests/dev: it uses the oracle, which only the test tree may)
sys.path[:0] = [os.path.join(ROOT, "bayes.js_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import amwg_ctypes as A, model_spec, oracle_lib, golden_io
name = sys.argv[1] if len(sys.argv) > 1 else "hier_small"
gl = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
gold = golden_io.load(name); case = gold["case"]; rec = gold["chains"][0]
spec = model_spec.spec_from_golden(gold, rec)
s = A.Sampler(spec, chains=2, seed=case["seed"], chain_offset=rec["chain"], group_local=gl, lanes_per_chain=64)
o = oracle_lib.OracleChain(spec, case["seed"], rec["chain"], lanes=64, group_local=bool(gl))
d = s.diag()
print("init lp gpu %r orc %r" % (float(d["log_post"][0]), o.log_post()))
for t in range(steps):
    s.burn(1); o.burn(1)
    d = s.diag(); gi = s.info(); oi = o.info()
    same_state = s.state()[:, 0].tobytes() == o.state().tobytes()
    same_u = int(d["uniforms"][0]) == o.uniforms()
    same_lp = np.float64(d["log_post"][0]).tobytes() == np.float64(o.log_post()).tobytes()
    same_acc = gi["accepts"][:, 0].tolist() == oi["accepts"].tolist()
    print("step %d state %s uniforms %s (%d vs %d) lp %s acc %s" % (t, same_state, same_u, int(d["uniforms"][0]), o.uniforms(), same_lp, same_acc))
    if not (same_state and same_u and same_lp and same_acc):
        print("gpu state", s.state()[:, 0]); print("orc state", o.state())
        print("gpu lp %r orc lp %r" % (float(d["log_post"][0]), o.log_post()))
        print("gpu acc", gi["accepts"][:, 0].tolist()); print("orc acc", oi["accepts"].tolist())
        print("gpu inb", gi["inbounds"][:, 0].tolist()); print("orc inb", oi["inbounds"].tolist())
        print("order gpu", d["named_order"][0].tolist(), "orc", o.named_order().tolist())
        s.burn(0)
        print("gpu lp after a 0-step launch (caches re-formed from the state): %r" % float(s.diag()["log_post"][0]))
        break
print("---- multi-step launches")
s2 = A.Sampler(spec, chains=5, seed=case["seed"], chain_offset=rec["chain"], group_local=gl, lanes_per_chain=64)
o2 = oracle_lib.OracleChain(spec, case["seed"], rec["chain"], lanes=64, group_local=bool(gl))
for n in (2, 3, 5, 10, 30, 50):
    s2.burn(n); o2.burn(n)
    d = s2.diag()
    print("burn(%d): state %s uniforms %d vs %d lp %s acc %s pls %s" % (n, s2.state()[:, 0].tobytes() == o2.state().tobytes(), int(d["uniforms"][0]), o2.uniforms(),
          np.float64(d["log_post"][0]).tobytes() == np.float64(o2.log_post()).tobytes(), s2.info()["accepts"][:, 0].tolist() == o2.info()["accepts"].tolist(),
          s2.info()["prop_log_scale"][:, 0].tobytes() == o2.info()["prop_log_scale"].tobytes()))
g = s2.sample(12, 3); w = o2.sample(12, 3)
print("sample(12,3): draws equal %s ; state %s" % (g[:, :, 0].tobytes() == np.ascontiguousarray(w).tobytes(), s2.state()[:, 0].tobytes() == o2.state().tobytes()))
if g[:, :, 0].tobytes() != np.ascontiguousarray(w).tobytes():
    print(g[:, :, 0]); print(w)
NL = int(sys.argv[4]) if len(sys.argv) > 4 else 2
NC = int(sys.argv[5]) if len(sys.argv) > 5 else 1
print("---- t single steps, then one %d-step launch, %d chains" % (NL, NC))
for t in range(0, 8):
    s3 = A.Sampler(spec, chains=NC, seed=case["seed"], chain_offset=rec["chain"], group_local=gl, lanes_per_chain=64)
    o3 = oracle_lib.OracleChain(spec, case["seed"], rec["chain"], lanes=64, group_local=bool(gl))
    for _ in range(t):
        s3.burn(1); o3.burn(1)
    o3.burn(1); order1 = o3.named_order().tolist(); acc1 = o3.info()["accepts"].copy(); u1 = o3.uniforms()
    o3.burn(1); order2 = o3.named_order().tolist()
    o3.burn(NL - 2)
    s3.burn(NL)
    ok = s3.state()[:, 0].tobytes() == o3.state().tobytes()
    print("t=%d orders %s %s  uniforms after first %d (mod 128 = %d) -> state %s uniforms %d vs %d" % (t, order1, order2, u1, u1 % 128, ok, int(s3.diag()["uniforms"][0]), o3.uniforms()))
    if not ok:
        print(" gpu", s3.state()[:, 0]); print(" orc", o3.state())
    s3.close()
