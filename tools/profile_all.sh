set -x
cd $GRAFT_REPO_ROOT
bash tools/profile.sh r05_cfg2 --steps 500 --warmup 1000 > /dev/null 2>&1; python tools/summarize_profile.py r05_cfg2 > /dev/null 2>&1
bash tools/profile.sh r05_cfg2full --full-evaluation --steps 300 --warmup 600 > /dev/null 2>&1; python tools/summarize_profile.py r05_cfg2full > /dev/null 2>&1
bash tools/profile.sh r05_cfg4 --workload cfg4 --weak --steps 300 --warmup 600 > /dev/null 2>&1; python tools/summarize_profile.py r05_cfg4 > /dev/null 2>&1
bash tools/profile.sh r05_cfg4full --workload cfg4 --weak --full-evaluation --steps 300 --warmup 600 > /dev/null 2>&1; python tools/summarize_profile.py r05_cfg4full > /dev/null 2>&1
bash tools/profile.sh r05_cfg4gl --workload cfg4 --weak --group-local --steps 300 --warmup 600 > /dev/null 2>&1; python tools/summarize_profile.py r05_cfg4gl > /dev/null 2>&1
bash tools/profile.sh r05_cfg5 --workload cfg5 --weak --steps 60 --warmup 60 --steps-per-launch 20 > /dev/null 2>&1; python tools/summarize_profile.py r05_cfg5 > /dev/null 2>&1
bash tools/profile.sh r05_cfg5full --workload cfg5 --weak --full-evaluation --steps 40 --warmup 40 --steps-per-launch 20 > /dev/null 2>&1; python tools/summarize_profile.py r05_cfg5full > /dev/null 2>&1
mkdir -p gpurun_out/r05_profiles; cp profiles/r05_cfg* gpurun_out/r05_profiles/
ls gpurun_out/r05_profiles | wc -l
python tools/flip_rate.py > gpurun_out/r05_flip_rate.json 2> gpurun_out/r05_flip_rate.log; tail -9 gpurun_out/r05_flip_rate.log
