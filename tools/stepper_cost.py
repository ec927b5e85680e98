"""Development tool (GPU box, under rocprofv3 --pmc): the cost of the stepper alone -- a model family with almost no data (one observation per
lane), so that nearly every instruction of an update is Philox / proposal / priors / butterfly / accept / adaptation.
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d out -o pmc --output-format csv -- python tools/stepper_cost.py hier_normal 64
prints the launch geometry; divide the counters of the 200-step launch by chains x 200 x P x (lanes / 64)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "bayes.js_amd"), os.path.join(ROOT, "tests")]
import amwg_ctypes as A, model_spec
fam, lanes = sys.argv[1], int(sys.argv[2])
n_obs = int(sys.argv[3]) if len(sys.argv) > 3 else lanes
chains = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
spec = model_spec.build_spec(fam, model_spec.make_data(fam, n_obs, 20260925, G=32, exp=A.lib().amwg_exp))
NS0 = int(os.environ.get("AMWG_STEPS", "200"))
gl = int(os.environ.get("AMWG_GL", "0"))
s = A.Sampler(spec, chains=chains, seed=1, lanes_per_chain=lanes, steps_per_launch=NS0, group_local=gl)
NS = int(os.environ.get("AMWG_STEPS", "200"))
s.burn(2 * NS)          # adapted
s.burn(NS)          # the launch to look at (the last one)
li = s.launch_info()
print("%sfamily %s n_obs %d chains %d P %d lanes %d block %d grid %d  kernel_ms %.3f  -> %.3f us per update-round" % ("[group-local] " if gl else "", fam, n_obs, chains, spec["P"], li["lanes_per_chain"], li["block_threads"], li["grid_blocks"], li["kernel_ms"], li["kernel_ms"] * 1e3 / (NS * spec["P"])))
s.close()
