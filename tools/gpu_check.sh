#!/bin/bash
# Runs on the GPU box (via gpurun): the -m gpu suite, then short benches of the four BASELINE configs + the single-chain README case.
#   tools/gpu_check.sh <tag> [pytest args]
TAG=${1:-r03}
shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -q "$@" 2>&1 | tail -40 > $OUT/${TAG}_pytest_gpu.log
tail -3 $OUT/${TAG}_pytest_gpu.log
for W in cfg4 cfg2 cfg5 cfg3; do
  python bench.py --workload $W --no-cpu-baseline --min-seconds 0.6 > $OUT/${TAG}_bench_$W.json 2> $OUT/${TAG}_bench_$W.err
  python - <<PY
import json
try:
    r = json.load(open("$OUT/${TAG}_bench_$W.json"))
    print("$W", "%.4g" % r["value"], "frac %.3f" % r["roofline"]["frac"], "lanes", r["config"]["lanes_per_chain"], "block", r["config"]["block_threads"])
except Exception as e:
    print("$W failed", e, open("$OUT/${TAG}_bench_$W.err").read()[-800:])
PY
done
python bench.py --workload cfg4 --chains-per-gpu 16384 --no-cpu-baseline --min-seconds 0.6 > $OUT/${TAG}_bench_cfg4_16384.json 2>/dev/null
python bench.py --workload readme --steps 20000 --warmup 2000 --thin 1 --no-cpu-baseline --min-seconds 0.5 > $OUT/${TAG}_bench_readme.json 2>/dev/null
python - <<PY
import json
for n in ("cfg4_16384", "readme"):
    try:
        r = json.load(open("$OUT/${TAG}_bench_%s.json" % n))
        print(n, "%.4g" % r["value"], "frac %.3f" % r["roofline"]["frac"], "lanes", r["config"]["lanes_per_chain"], "block", r["config"]["block_threads"])
    except Exception as e:
        print(n, "failed", e)
PY
