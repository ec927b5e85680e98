#!/usr/bin/env python3
"""Development tool (GPU box): per-update instruction / wait anatomy of one step-kernel configuration from rocprofv3 PMC counters.

    python tools/pmc_probe.py <tag> <family> <lanes> <n_obs> <chains> [<family> <lanes> <n_obs> <chains> ...]

For every configuration runs tools/stepper_cost.py (400 + 200 burn-in steps, then the 200-step launch that is looked at) under
`rocprofv3 --kernel-trace --pmc ...` in two passes (8 SQ counters each; counters in their own runs, no other trace domains) and prints,
per parameter update and wavefront: VALU / SALU / LDS / SMEM instructions, the share of wave cycles with a VALU instruction in flight,
SQ_WAIT_ANY and SQ_WAIT_INST_ANY, and the time per update-round.  n_obs = lanes gives the stepper alone (one observation per lane).
Writes gpurun_out/<tag>_pmc.json (copy what matters into profiles/)."""
import csv, glob, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY"],
          ["SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAVES", "SQ_ACTIVE_INST_SCA", "SQ_INSTS_SMEM", "GRBM_GUI_ACTIVE"]]


if os.environ.get("AMWG_PMC_PASSES"):          # e.g. "TCC_HIT_sum,TCC_MISS_sum;SQ_INSTS_VALU,SQ_WAVE_CYCLES": other counters, raw sums only
    PASSES = [p.split(",") for p in os.environ["AMWG_PMC_PASSES"].split(";")]


def probe(tag, fam, lanes, n_obs, chains, idx):
    out = {}
    geom = None
    for pi, counters in enumerate(PASSES):
        d = "/tmp/pmc_%s_%d_%d" % (tag, idx, pi)
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "-o", "pmc", "--",
               sys.executable, os.path.join(ROOT, "tools", "stepper_cost.py"), fam, str(lanes), str(n_obs), str(chains)]
        p = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
        m = re.search(r"family.*", p.stdout)
        if m:
            geom = m.group(0)
        files = glob.glob(os.path.join(d, "**", "pmc_counter_collection.csv"), recursive=True)
        if not files:
            print("no counters for", fam, lanes, p.stderr[-500:])
            continue
        rows = [r for r in csv.DictReader(open(files[0])) if "amwg_step_kernel" in r["Kernel_Name"] or "amwg_user_step" in r["Kernel_Name"]]
        if not rows:
            continue
        last = max(int(r["Dispatch_Id"]) for r in rows)          # the last launch = the adapted 200-step one
        for r in rows:
            if int(r["Dispatch_Id"]) == last:
                out[r["Counter_Name"]] = out.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                out["_kernel"] = r["Kernel_Name"]
                out["_vgpr"], out["_sgpr"], out["_scratch"], out["_lds"] = r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("Scratch_Size"), r.get("LDS_Block_Size")
    return geom, out


def main():
    tag, args = sys.argv[1], sys.argv[2:]
    res = []
    for i in range(0, len(args), 4):
        fam, lanes, n_obs, chains = args[i], int(args[i + 1]), int(args[i + 2]), int(args[i + 3])
        geom, c = probe(tag, fam, lanes, n_obs, chains, i // 4)
        P = {"normal": 2, "beta_bern": 1, "hier_normal": 34, "pois_glm": 9}[fam]
        waves_per_chain = max(1.0, lanes / 64.0)
        upd = chains * 200 * P * waves_per_chain if lanes >= 64 else (chains * lanes / 64.0) * 200 * P     # update-rounds of a WAVE
        g = lambda k: c.get(k, float("nan"))
        row = {"config": {"family": fam, "lanes": lanes, "n_obs": n_obs, "chains": chains}, "geometry": geom, "kernel": c.get("_kernel"),
               "vgpr": c.get("_vgpr"), "sgpr": c.get("_sgpr"), "scratch": c.get("_scratch"),
               "valu_per_update": g("SQ_INSTS_VALU") / upd, "salu_per_update": g("SQ_INSTS_SALU") / upd, "lds_per_update": g("SQ_INSTS_LDS") / upd,
               "smem_per_update": g("SQ_INSTS_SMEM") / upd,
               "valu_active_share_of_wave_cycles": g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None,
               "wait_any_share": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), "wait_inst_any_share": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
               "wait_inst_lds_share": g("SQ_WAIT_INST_LDS") / g("SQ_WAVE_CYCLES"), "active_inst_lds_share": g("SQ_ACTIVE_INST_LDS") / g("SQ_WAVE_CYCLES"),
               "wave_cycles_per_update": g("SQ_WAVE_CYCLES") / upd, "busy_cycles": g("SQ_BUSY_CYCLES"), "lds_bank_conflict": g("SQ_LDS_BANK_CONFLICT"),
               "raw": {k: v for k, v in c.items() if not k.startswith("_")}}
        res.append(row)
        print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in row.items() if k != "raw" or os.environ.get("AMWG_PMC_PASSES")}))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", tag + "_pmc.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
