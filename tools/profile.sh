#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of one bench command.
#   tools/profile.sh <tag> [bench.py args...]        e.g.  tools/profile.sh r02_cfg4 --workload cfg4 --steps 300 --warmup 600
# Outputs under gpurun_out/prof_<tag>/ ; tools/summarize_profile.py <tag> condenses them into profiles/.
# Counters are collected in their own passes with --kernel-trace only (MI355X_MICROARCH.md: SQ 8 slots, TCC FETCH_SIZE and
# WRITE_SIZE do not fit one pass, GRBM independent); every pass re-runs the same command.
TAG=${1:-r03}
shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="${@:---steps 500 --warmup 1000}"   # (cfg4 / cfg5: add --weak to profile the per-GPU shard of the 8-GPU job, 2 048 / 8 192 chains)
CMD="python $R/bench.py --no-cpu-baseline --single-region --no-other-configs --no-parity $ARGS"
echo "$CMD" > $OUT/command.txt
# (bench.py's stdout is the <= 4 KB line; the whole record -- what tools/summarize_profile.py reads -- is the detail file)
AMWG_BENCH_DETAIL=$OUT/bench_under_trace.json rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/bench_line.json 2> $OUT/trace_err.log
export AMWG_BENCH_DETAIL=/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $CMD > /dev/null 2> $OUT/pmc_fetch_err.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $CMD > /dev/null 2> $OUT/pmc_write_err.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o pmc -- $CMD > /dev/null 2> $OUT/pmc_sq_err.log
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/pmc_sq2 -o pmc -- $CMD > /dev/null 2> $OUT/pmc_sq2_err.log
find $OUT -name "*.csv" | head -30
# keep only small files
find $OUT -size +2M -delete
