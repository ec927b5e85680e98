"""Condenses gpurun_out/prof_<tag>/ (written by tools/profile.sh on the GPU box) into the small
files kept under profiles/: the rocprofv3 --kernel-trace --stats table, per-launch PMC means for the
step kernel, and the HBM traffic figure bench.py reports (MI355X_MICROARCH.md §HBM: FETCH_SIZE and
WRITE_SIZE are in KB, collected in separate passes; on gfx950 FETCH_SIZE counts 64 B per 128-B
request for wide coalesced reads, so the read side is doubled)."""
import collections, csv, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", "prof_" + tag), os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_under_trace.json"), os.path.join(dst, tag + "_bench_under_trace.json"))
bench = json.load(open(os.path.join(src, "bench_under_trace.json")))
# the kernel the bench line is about (the parity gate of bench.py also launches one-lane kernels: not those)
import re
m = re.match(r"amwg_step_kernel<(\w+),(\d+)>", bench["roofline"]["kernel"])
is_bench_kernel = lambda name: ("amwg_step_kernel" in name and re.search(r"%s,\s*%s>" % (m.group(1), m.group(2)), name) is not None)
pmc = {}
meta = {}
for name in ("pmc_fetch", "pmc_write", "pmc_sq"):
    f = os.path.join(src, name, "pmc_counter_collection.csv")
    if not os.path.exists(f):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if is_bench_kernel(r["Kernel_Name"]):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
    for k, v in agg.items():
        pmc[k] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
stats = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))))
step = [r for r in stats if is_bench_kernel(r["Name"])][0]
traffic = None
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    traffic = (2.0 * pmc["FETCH_SIZE"]["mean"] + pmc["WRITE_SIZE"]["mean"]) * 1024.0
out = {"tag": tag, "command": "python bench.py --no-cpu-baseline --steps 500 --warmup 1000 (100 steps per launch)",
       "kernel": step["Name"], "rocprof_calls": int(step["Calls"]), "rocprof_avg_launch_ms": float(step["AverageNs"]) / 1e6,
       "bench_launch_ms": bench["roofline"]["launch_ms"], "steps_per_launch": bench["config"]["steps_per_launch"],
       "chains": bench["config"]["chains_per_gpu"], "kernel_resources": meta, "pmc_per_launch": pmc,
       "hbm_traffic_bytes_per_launch": traffic,
       "hbm_traffic_formula": "(2*FETCH_SIZE + WRITE_SIZE) KB -> bytes; read side doubled per MI355X_MICROARCH.md (gfx950 FETCH_SIZE = RDREQ*64B for 128-B requests)",
       "algorithmic_bytes_per_launch": bench["config"]["chains_per_gpu"] * bench["config"]["steps_per_launch"] * bench["config"]["components"] * bench["roofline"]["algorithmic_bytes_per_update"]}
json.dump(out, open(os.path.join(dst, tag + "_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:1800])
