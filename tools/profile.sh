#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of the bench command.
# Outputs under gpurun_out/prof_<tag>/ ; summaries are copied by hand into profiles/.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --no-cpu-baseline --steps 500 --warmup 1000"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/bench_under_trace.json 2> $OUT/trace_err.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $CMD > /dev/null 2> $OUT/pmc_fetch_err.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $CMD > /dev/null 2> $OUT/pmc_write_err.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d $OUT/pmc_sq -o pmc -- $CMD > /dev/null 2> $OUT/pmc_sq_err.log
find $OUT -name "*.csv" | head -30
# keep only small files
find $OUT -size +2M -delete
