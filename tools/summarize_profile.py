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
# ("amwg_step_kernel_cert<NormalModel,1,256> (certified decisions)" -> function name, template arguments; the kernels that decide from certified values are
# functions of their own, amwg_step_kernel_cert / amwg_sweep_kernel_cert: a profile that also holds the plain kernel -- bench.py's full_evaluation side measurement --
# must not mix the two)
m = re.match(r"(amwg_step_kernel(?:_cert)?)<(\w+),(\d+)(?:,(\d+))?>", bench["roofline"]["kernel"]) or re.match(r"(amwg_(?:sweep|gl)_kernel(?:_cert)?)<(\w+),(\d+)>", bench["roofline"]["kernel"])
workload = re.search(r"--workload (\w+)", open(os.path.join(src, "command.txt")).read() if os.path.exists(os.path.join(src, "command.txt")) else "")
workload = workload.group(1) if workload else "cfg2"
is_bench_kernel = lambda name: (m.group(1) + "<" in name and re.search(r"%s,\s*%s(,\s*\d+)?>" % (m.group(2), m.group(3)), name) is not None)
if "--group-local" in (open(os.path.join(src, "command.txt")).read() if os.path.exists(os.path.join(src, "command.txt")) else ""):
    is_bench_kernel = lambda name: "amwg_gl_kernel" in name      # the group-local evaluation is its own kernel since round 4 (csrc/amwg_gl.h)
elif bench["roofline"]["kernel"].startswith("amwg_sweep_kernel"):
    is_bench_kernel = lambda name: m.group(1) + "<" in name   # the hierarchical family's row layout (the certified sweep kernel, or the lane-order one)
pmc = {}
meta = {}
for name in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    f = os.path.join(src, name, "pmc_counter_collection.csv")
    if not os.path.exists(f):
        continue
    agg = collections.defaultdict(list)
    rows = [r for r in csv.DictReader(open(f)) if is_bench_kernel(r["Kernel_Name"])]
    # the parity block of bench.py launches the same kernel on ONE chain (tiny grids): only the full-size launches are the bench's
    full = max((int(r["Grid_Size"]) for r in rows), default=0)
    for r in rows:
        if int(r["Grid_Size"]) == full:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
    for k, v in agg.items():
        pmc[k] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
stats = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))))
step = [r for r in stats if is_bench_kernel(r["Name"])][0]
# per-launch durations of the full-size launches from the kernel trace (the stats table averages the one-chain parity launches in)
tr = [r for r in csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_trace.csv"))) if is_bench_kernel(r["Kernel_Name"])]
full_grid = max(int(r["Grid_Size_X"]) for r in tr)
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in tr if int(r["Grid_Size_X"]) == full_grid]
step = dict(step, Calls=len(dur), AverageNs=sum(dur) / len(dur), MinNs=min(dur), MaxNs=max(dur))
traffic = None
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    traffic = (2.0 * pmc["FETCH_SIZE"]["mean"] + pmc["WRITE_SIZE"]["mean"]) * 1024.0
derived = {}
g = lambda k: pmc[k]["mean"] if k in pmc else None
launch_s = float(step["AverageNs"]) * 1e-9
if g("GRBM_GUI_ACTIVE"):
    derived["effective_clock_ghz"] = g("GRBM_GUI_ACTIVE") / 8.0 / launch_s / 1e9      # MI355X_MICROARCH.md "DVFS give-back"; the counter is summed over the 8 XCDs
    derived["effective_clock_note"] = "GRBM_GUI_ACTIVE / 8 XCDs / rocprofv3 launch duration (of the PMC pass average)"
if g("SQ_WAVE_CYCLES") and g("SQ_ACTIVE_INST_VALU"):
    derived["valu_active_share_of_wave_cycles"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")
    derived["wait_inst_any_share_of_wave_cycles"] = (g("SQ_WAIT_INST_ANY") or 0) / g("SQ_WAVE_CYCLES")
    derived["wait_any_share_of_wave_cycles"] = (g("SQ_WAIT_ANY") or 0) / g("SQ_WAVE_CYCLES")
# How busy the vector pipe is, from counts alone (round-4 review): a wave64 VALU instruction occupies its SIMD's 16-lane pipe for 4 cycles (fp64 runs at full
# rate on this chip); a transcendental (v_rcp_f64 ...) for ~16.5 (tools/ubench/valu_rates.hip), i.e. 12.5 more.  SIMD-cycles = 1024 SIMDs x launch time x the
# effective clock.  TRANS_PER_UNIT: transcendental instructions per observation (static count in the kernel's pass; PoisGlm: the two v_rcp_f64 of exp and log).
TRANS_PER_OBS = {"cfg5": 2.0, "cfg2": 0.0, "cfg4": 0.0, "cfg3": 0.0}
if g("SQ_INSTS_VALU") and derived.get("effective_clock_ghz"):
    simd_cycles = 1024.0 * launch_s * derived["effective_clock_ghz"] * 1e9
    n_obs_lane_iterations = (bench["roofline"].get("updates_per_launch") or 0) * bench["config"]["n_obs"] / 64.0      # per-wave passes over an observation
    cmd_text = open(os.path.join(src, "command.txt")).read() if os.path.exists(os.path.join(src, "command.txt")) else ""
    # (cfg5: the two reciprocals belong to the expression's exp / log; the certified pass -- the default since round 5 -- has none)
    trans = (TRANS_PER_OBS.get(workload, 0.0) if ("--full-evaluation" in cmd_text or workload != "cfg5") else 0.0) * n_obs_lane_iterations
    derived["valu_pipe_busy"] = (g("SQ_INSTS_VALU") * 4.0 + trans * 12.5) / simd_cycles
    derived["valu_pipe_busy_without_transcendentals"] = g("SQ_INSTS_VALU") * 4.0 / simd_cycles
    derived["valu_pipe_busy_note"] = "(SQ_INSTS_VALU x 4 cycles + %g transcendental instructions x 12.5 extra cycles) / (1024 SIMDs x launch time x effective clock)" % trans
if g("SQ_BUSY_CYCLES") and g("SQ_ACTIVE_INST_VALU"):
    # SQ_ACTIVE_INST_VALU counts quad-cycles summed over waves; SQ_BUSY_CYCLES counts cycles per SE/XCC: report the raw ratio only
    derived["active_inst_valu_per_busy_cycle"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_BUSY_CYCLES")
units = bench["roofline"].get("updates_per_launch") or (bench["config"]["chains_per_gpu"] * bench["config"]["steps_per_launch"] * bench["config"]["components"])
if g("SQ_INSTS_VALU"):
    derived["valu_instructions_per_update_per_wave64"] = g("SQ_INSTS_VALU") / (units * bench["config"]["lanes_per_chain"] / 64.0)
    derived["valu_instructions_per_observation_lane"] = g("SQ_INSTS_VALU") * 64.0 / (units * bench["config"]["n_obs"])
_ver = (bench.get("library") or {}).get("version", "")
_kid = re.search(r"kernels ([0-9a-f]{12})", _ver)
out = {"tag": tag, "workload": workload, "library_version": _ver, "kernel_id": _kid.group(1) if _kid else None, "command": open(os.path.join(src, "command.txt")).read().strip() if os.path.exists(os.path.join(src, "command.txt")) else "python bench.py --no-cpu-baseline --steps 500 --warmup 1000 (100 steps per launch)",
       "derived": derived,
       "kernel": step["Name"], "rocprof_calls": int(step["Calls"]), "rocprof_avg_launch_ms": float(step["AverageNs"]) / 1e6,
       "rocprof_min_launch_ms": step["MinNs"] / 1e6, "rocprof_max_launch_ms": step["MaxNs"] / 1e6,
       "rocprof_note": "full-size launches only (grid %d); the kernel_stats.csv row also averages the one-chain launches of bench.py's parity block" % full_grid,
       "bench_launch_ms": bench["roofline"]["launch_ms"], "steps_per_launch": bench["config"]["steps_per_launch"],
       "chains": bench["config"]["chains_per_gpu"], "components": bench["config"].get("components"), "kernel_resources": meta, "pmc_per_launch": pmc,
       "hbm_traffic_bytes_per_launch": traffic,
       "hbm_traffic_formula": "(2*FETCH_SIZE + WRITE_SIZE) KB -> bytes; read side doubled per MI355X_MICROARCH.md (gfx950 FETCH_SIZE = RDREQ*64B for 128-B requests)",
       "algorithmic_bytes_per_launch": bench["config"]["chains_per_gpu"] * bench["config"]["steps_per_launch"] * bench["config"]["components"] * bench["roofline"]["algorithmic_bytes_per_update"]}
json.dump(out, open(os.path.join(dst, tag + "_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:1800])
