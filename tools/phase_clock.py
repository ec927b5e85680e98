#!/usr/bin/env python3
"""tools/phase_clock.py -- where does a latency-bound stepper spend its time?  (development tool; needs a GPU)

Runs cfg4's certified sweep kernel (and, with --cfg 2 / 5, the other benched kernels) from a build with -DAMWG_X_PHASES (tools/build_variant.sh phases -DAMWG_X_PHASES):
the step loop reads the shader clock at its phase boundaries and sums the cycles per phase over all wavefronts (csrc/amwg_kernel.h AMWG_PHASE).  Prints cycles per step
and wavefront by phase.    python tools/phase_clock.py [--cfg 4]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("AMWG_LIB", os.path.join(ROOT, "build", "phases", "libamwg.so"))
for p in (ROOT, os.path.join(ROOT, "bayes.js_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

import amwg_ctypes as A  # noqa: E402
import model_spec  # noqa: E402

NAMES = {0: "record + shuffle of the named parameters", 1: "first look-ahead (next_comp + prefetch)", 2: "sweep: window walk + proposals", 3: "sweep: sweep_approx (the S2 pass)",
         4: "sweep: local differences + M butterfly + bound", 5: "sweep: exp_v8", 6: "sweep: test, ballots, commit, value butterfly, counters / adaptation", 7: "sweep: look-ahead of the next slot",
         8: "slot: proposal (rnorm) + bounds + accept uniform", 9: "slot: look-ahead of the next slot (incl. theta's shuffle)", 10: "slot: log_post_approx", 11: "slot: exp + certified test + commit",
         12: "slot: adaptation", 13: "sweep: entry + second half of the window (Philox, rejection flags)", 14: "slot entry", 15: "loop back edge"}
cfg = int(sys.argv[sys.argv.index("--cfg") + 1]) if "--cfg" in sys.argv else 4
if cfg == 4:
    d = model_spec.make_data("hier_normal", 10000, 20260925, G=32)
    spec, chains, lanes, steps = model_spec.build_spec("hier_normal", d), 2048, 64, 200
elif cfg == 40:      # BASELINE configs[3] as a plain closure, translated: the certified row plan (amwg_user_sweep_cert)
    import user_host
    src, arrays, meta = user_host.translated("bench_hier")
    inf = float("inf")
    params = [{"type": "real", "len": 32, "top": 32, "multidim": 1, "lower": -inf, "upper": inf}, {"type": "real", "len": 1, "top": 1, "multidim": 0, "lower": -inf, "upper": inf},
              {"type": "real", "len": 1, "top": 1, "multidim": 0, "lower": 0.0, "upper": inf}]
    spec = {"user": user_host.user_spec_part(src, arrays, meta), "params": params, "P": 34, "init": [0.5] * 32 + [0.5, 1.0], "comp_opts": [dict(model_spec.DEFAULT_OPT) for _ in range(34)]}
    chains, lanes, steps = 2048, 64, 200
elif cfg == 1:      # README.md:18-43 as is: ONE chain on the ten heights (the latency of one wavefront's dependent chain)
    spec, chains, lanes, steps = model_spec.build_spec("normal", {"x": np.array([183, 192, 182, 183, 177, 185, 188, 188, 182, 185], dtype=np.float64)}), 1, 64, 2000
elif cfg == 2:
    spec, chains, lanes, steps = model_spec.build_spec("normal", model_spec.make_data("normal", 10000, 20260925)), 65536, 1, 100
else:
    spec, chains, lanes, steps = model_spec.build_spec("pois_glm", model_spec.make_data("pois_glm", 50000, 20260925)), 8192, 16, 20
s = A.Sampler(spec, chains=chains, seed=1, lanes_per_chain=lanes, steps_per_launch=100)
s.burn(600 if cfg != 5 else 40)
L = A.lib()
L.amwg_audit_fetch.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int32]
hist = np.zeros(128, dtype=np.uint64)
L.amwg_audit_fetch(s.h, None, hist.ctypes.data_as(C.POINTER(C.c_uint64)), 1)
s.burn(steps)
ms = s.launch_info()["kernel_ms"]
L.amwg_audit_fetch(s.h, None, hist.ctypes.data_as(C.POINTER(C.c_uint64)), 0)
waves = chains * lanes // 64 if lanes >= 64 else (chains * lanes + 63) // 64
ph = hist[64:80].astype(np.float64) / (waves * steps)
print("%s  %d chains, %d steps: %.3f ms per 100 steps; cycles per step and wavefront (shader clock), total %.0f" % (s.launch_info()["kernel"], chains, steps, ms * 100 / steps, ph.sum()))
for i in range(16):
    if ph[i] > 0:
        print("  %2d  %-72s %9.0f  %5.1f %%" % (i, NAMES.get(i, "?"), ph[i], 100 * ph[i] / ph.sum()))
