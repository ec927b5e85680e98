#!/bin/bash
# GPU box: runs a command under rocprofv3 PMC passes and prints the per-kernel means of a few counters
#   tools/pmc_cmd.sh <tag> <command ...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/a -o pmc -- "$@" > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/b -o pmc -- "$@" > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("a", "b"):
    f = glob.glob("$OUT/%s/**/pmc_counter_collection.csv" % d, recursive=True)
    if not f: print("no counters in", d); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        n = max(len(v) for v in c.values())
        if n < 1: continue
        print(k, "launches", n, {cn: "%.4g" % (sum(v) / len(v)) for cn, v in c.items()})
PY
find $OUT -size +1M -delete
