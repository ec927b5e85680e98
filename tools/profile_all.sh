# Round-6 evidence of the FINAL kernels, in one gpurun call (bash tools/profile_all.sh): rocprofv3 kernel-trace + PMC passes of every benched kernel (tools/profile.sh ->
# profiles/r06_<cfg>_summary.json, tagged with the kernel id), the campaigns that count decisions, the bound audit, and the bench lines.  Copies land in gpurun_out/r06/.
set -x
cd $GRAFT_REPO_ROOT
T=r06
bash tools/profile.sh ${T}_cfg2 --steps 500 --warmup 1000 > /dev/null 2>&1; python tools/summarize_profile.py ${T}_cfg2 > /dev/null 2>&1
bash tools/profile.sh ${T}_cfg2full --full-evaluation --steps 300 --warmup 600 > /dev/null 2>&1; python tools/summarize_profile.py ${T}_cfg2full > /dev/null 2>&1
bash tools/profile.sh ${T}_cfg3 --workload cfg3 --steps 300 --warmup 600 > /dev/null 2>&1; python tools/summarize_profile.py ${T}_cfg3 > /dev/null 2>&1
bash tools/profile.sh ${T}_cfg4 --workload cfg4 --weak --steps 300 --warmup 600 > /dev/null 2>&1; python tools/summarize_profile.py ${T}_cfg4 > /dev/null 2>&1
bash tools/profile.sh ${T}_cfg4full --workload cfg4 --weak --full-evaluation --steps 300 --warmup 600 > /dev/null 2>&1; python tools/summarize_profile.py ${T}_cfg4full > /dev/null 2>&1
bash tools/profile.sh ${T}_cfg5 --workload cfg5 --weak --steps 60 --warmup 60 --steps-per-launch 20 > /dev/null 2>&1; python tools/summarize_profile.py ${T}_cfg5 > /dev/null 2>&1
bash tools/profile.sh ${T}_cfg5full --workload cfg5 --weak --full-evaluation --steps 40 --warmup 40 --steps-per-launch 20 > /dev/null 2>&1; python tools/summarize_profile.py ${T}_cfg5full > /dev/null 2>&1
mkdir -p gpurun_out/$T; cp profiles/${T}_cfg* gpurun_out/$T/
python tools/bound_audit.py --shrink --out gpurun_out/$T/${T}_bound_audit.json > gpurun_out/$T/${T}_bound_audit.log 2>&1; tail -3 gpurun_out/$T/${T}_bound_audit.log
python tools/flip_rate.py > gpurun_out/$T/${T}_flip_rate.json 2> gpurun_out/$T/${T}_flip_rate.log; tail -5 gpurun_out/$T/${T}_flip_rate.log
python tools/certified_campaign.py --scale 0.5 > gpurun_out/$T/${T}_certified_campaign.json 2> gpurun_out/$T/${T}_certified_campaign.log; tail -3 gpurun_out/$T/${T}_certified_campaign.log
cp gpurun_out/$T/${T}_flip_rate.json gpurun_out/$T/${T}_cfg*_summary.json profiles/ 2>/dev/null      # (so that the bench lines below find evidence of THESE kernels)
[ -f build/phases/libamwg.so ] || bash tools/build_variant.sh phases -DAMWG_X_PHASES > /dev/null 2>&1      # (the phase-clock development build; the box has hipcc)
python tools/phase_clock.py --cfg 4 > gpurun_out/$T/${T}_phase_clock.txt 2>&1; python tools/phase_clock.py --cfg 2 >> gpurun_out/$T/${T}_phase_clock.txt 2>&1; python tools/phase_clock.py --cfg 5 >> gpurun_out/$T/${T}_phase_clock.txt 2>&1; python tools/phase_clock.py --cfg 40 >> gpurun_out/$T/${T}_phase_clock.txt 2>&1; python tools/phase_clock.py --cfg 1 >> gpurun_out/$T/${T}_phase_clock.txt 2>&1
python bench.py > gpurun_out/$T/${T}_bench_default.json 2> /dev/null; cp bench_detail.json gpurun_out/$T/${T}_bench_default_detail.json
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$T/${T}_bench_driver_line.json 2> /dev/null; cp bench_detail.json gpurun_out/$T/${T}_bench_driver_line_detail.json
python bench.py --workload cfg4 --weak --steps 300 --warmup 600 --no-cpu-baseline > gpurun_out/$T/${T}_bench_cfg4_line.json 2> /dev/null
python bench.py --workload cfg5 --weak --steps 60 --warmup 60 --steps-per-launch 20 --no-cpu-baseline > gpurun_out/$T/${T}_bench_cfg5_line.json 2> /dev/null
python bench.py --workload cfg3 --steps 300 --warmup 600 --no-cpu-baseline > gpurun_out/$T/${T}_bench_cfg3_line.json 2> /dev/null
AMWG_TIMING=1 node bench/js_e2e.js > gpurun_out/$T/${T}_js_e2e.json 2> gpurun_out/$T/${T}_ctor_timing.txt
tools/ubench/pinned_copy 1024 > gpurun_out/$T/${T}_pinned_copy.txt 2>&1
ls gpurun_out/$T | wc -l
