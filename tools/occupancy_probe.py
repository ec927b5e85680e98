"""Development tool (GPU box): the hierarchical kernel (cfg4 geometry: 2 048 chains, a wavefront per chain) with 512-thread workgroups (one per CU: two waves per SIMD)
and with 256-thread ones (the staged data leaves room for one per CU: ONE wave per SIMD, two rounds of workgroups) -- how much of the time per update round does a second
wave per SIMD hide?  (End of round 3: 6.95 vs 9.21 us, a factor 1.32 where a latency-bound kernel would show 2 and a lone wave's slower issue alone 1.2: issue bound.)
    python tools/occupancy_probe.py"""
import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [os.path.join(ROOT, "bayes.js_amd"), os.path.join(ROOT, "tests")]
import amwg_ctypes as A, model_spec
for n_obs in (64, 10000):
    spec = model_spec.build_spec("hier_normal", model_spec.make_data("hier_normal", n_obs, 20260925, G=32, exp=A.lib().amwg_exp))
    for bt in (512, 256):
        s = A.Sampler(spec, chains=2048, seed=1, lanes_per_chain=64, block_threads=bt, steps_per_launch=100)
        s.burn(200); s.burn(100)
        li = s.launch_info()
        print("n_obs %d block %d grid %d lds %d: kernel_ms %.3f -> %.3f us per update-round (2048 chains)" % (n_obs, li["block_threads"], li["grid_blocks"], li["lds_bytes"], li["kernel_ms"], li["kernel_ms"] * 1e3 / (100 * spec["P"])))
        s.close()
