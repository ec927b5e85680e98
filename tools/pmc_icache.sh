#!/bin/bash
# GPU box: instruction-cache counters of whatever kernels a command launches (development tool; round 6: is a 70-90 KB step loop fetch-bound?)
#   tools/pmc_icache.sh <command ...>      -> per kernel: SQC_ICACHE_REQ / HITS / MISSES per launch (max-grid launches), SQ_IFETCH, wave cycles
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pmc_ic; rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d /tmp/pmc_ic -o pmc -- "$@" > /tmp/pmc_ic.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc_ic/**/pmc_counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
grid = collections.defaultdict(int)
rows = list(csv.DictReader(open(f[0]))) if f else []
for r in rows:
    if "amwg" in r["Kernel_Name"]: grid[r["Kernel_Name"]] = max(grid[r["Kernel_Name"]], int(r["Grid_Size"]))
for r in rows:
    if "amwg" in r["Kernel_Name"] and int(r["Grid_Size"]) == grid[r["Kernel_Name"]]: agg[r["Kernel_Name"][:72]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    last = {n: v[-1] for n, v in c.items()}
    req = last.get("SQC_ICACHE_REQ", 0) or 1
    print(k, "| launches", len(next(iter(c.values()))), "| last launch: icache req %.4g hits %.4g misses %.4g (%.2f %%) dup %.4g | ifetch %.4g | valu %.4g | wave cycles %.4g wait_inst_any %.3f" % (
        last.get("SQC_ICACHE_REQ", 0), last.get("SQC_ICACHE_HITS", 0), last.get("SQC_ICACHE_MISSES", 0), 100.0 * last.get("SQC_ICACHE_MISSES", 0) / req, last.get("SQC_ICACHE_MISSES_DUPLICATE", 0),
        last.get("SQ_IFETCH", 0), last.get("SQ_INSTS_VALU", 0), last.get("SQ_WAVE_CYCLES", 0), last.get("SQ_WAIT_INST_ANY", 0) / (last.get("SQ_WAVE_CYCLES", 0) or 1)))
PY
