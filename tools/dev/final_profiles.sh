#!/bin/bash
# (GPU box) the round's closing measurements: the five rocprofv3 profiles, the default bench line, the driver's short line, the in-process group path
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
TAG=${1:-r04}
tools/profile.sh ${TAG}_cfg2 > /dev/null 2>&1
tools/profile.sh ${TAG}_cfg4 --workload cfg4 --weak --steps 300 --warmup 600 > /dev/null 2>&1
tools/profile.sh ${TAG}_cfg4full --workload cfg4 --weak --full-evaluation --steps 300 --warmup 600 > /dev/null 2>&1
tools/profile.sh ${TAG}_cfg4gl --workload cfg4 --weak --group-local --steps 300 --warmup 600 > /dev/null 2>&1
tools/profile.sh ${TAG}_cfg5 --workload cfg5 --weak --steps 40 --warmup 40 --steps-per-launch 20 > /dev/null 2>&1
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
python bench.py --steps 20 --warmup 5 --no-other-configs --no-js > gpurun_out/${TAG}_bench_driver_line.json 2>/dev/null
python bench.py --inproc --gpus 1 --workload cfg4 --strong > gpurun_out/${TAG}_inproc_cfg4_strong.json 2>/dev/null
python bench.py --workload cfg4 --weak --chains-per-gpu 16384 --no-cpu-baseline --no-other-configs --no-parity > gpurun_out/${TAG}_bench_cfg4_16384.json 2>/dev/null
ls -la gpurun_out/${TAG}_*.json
python - <<PY
import json
for n in ("bench_default", "bench_driver_line", "inproc_cfg4_strong", "bench_cfg4_16384"):
    try:
        r = json.load(open("gpurun_out/${TAG}_%s.json" % n))
        print(n, "%.4g" % r["value"], "frac %.3f" % r["roofline"]["frac"], r["roofline"].get("traffic"), r["roofline"].get("traffic_refused"))
        for k, o in (r.get("other_configs") or {}).items():
            print("   ", k, "%.4g" % o.get("value", 0), o.get("roofline", {}).get("frac"), o.get("error"))
    except Exception as e:
        print(n, "failed", e)
PY
