#!/usr/bin/env python3
"""development aid (GPU box): times and counts every build/cut/libamwg_cut<n>.so on cfg4 group-local (tools/dev/gl_cuts.sh)."""
import csv, glob, json, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out_root = os.path.join(R, "gpurun_out", "gl_cuts")
os.makedirs(out_root, exist_ok=True)
variants = sys.argv[1:] or [os.path.basename(f)[len("libamwg_cut"):-3] for f in sorted(glob.glob(os.path.join(R, "build", "cut", "libamwg_cut*.so")))]
extra = os.environ.get("CUT_ARGS", "--workload cfg4 --group-local").split()
rows = []
for v in variants:
    lib = os.path.join(R, "build", "cut", "libamwg_cut%s.so" % v)
    d = os.path.join(out_root, "v" + v)
    env = dict(os.environ, AMWG_LIB=lib, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_INSTS_LDS", "SQ_WAIT_INST_ANY", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
           sys.executable, os.path.join(R, "bench.py")] + extra + ["--steps", "200", "--warmup", "200", "--no-parity", "--no-cpu-baseline", "--no-js", "--single-region", "--no-other-configs"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd="/tmp")
    f = glob.glob(os.path.join(d, "**", "pmc_counter_collection.csv"), recursive=True)
    agg, ms = {}, None
    try:
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
        b = json.loads(line)
        ms = b["roofline"]["launch_ms"]
        upl = b["roofline"]["updates_per_launch"]
    except Exception as e:
        print(v, "bench failed", repr(e), p.stderr[-500:]); continue
    if f:
        rs = [r for r in csv.DictReader(open(f[0])) if "amwg_gl_kernel" in r["Kernel_Name"] or "amwg_step_kernel" in r["Kernel_Name"]]
        full = max(int(r["Grid_Size"]) for r in rs)
        for r in rs:
            if int(r["Grid_Size"]) == full: agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    g = lambda k: sum(agg[k]) / len(agg[k]) if k in agg else float("nan")
    waves_steps = b["config"]["chains_per_gpu"] * b["config"]["steps_per_launch"]
    rows.append((v, ms, g("SQ_INSTS_VALU") / waves_steps, g("SQ_INSTS_SALU") / waves_steps, g("SQ_INSTS_LDS") / waves_steps, g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAVE_CYCLES") * 4 / waves_steps))
    print("cut %-3s launch %.3f ms (under pmc)  per wave-step: VALU %.0f SALU %.0f LDS %.0f  wait_any %.3f wait_inst %.3f  wave-cycles %.0f" % rows[-1], flush=True)
json.dump(rows, open(os.path.join(out_root, "table.json"), "w"))
