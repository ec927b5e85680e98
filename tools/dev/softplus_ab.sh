#!/bin/bash
# (GPU box) A/B of the fused softplus: translated logistic closures with log1p_exp_v8 vs the two calls it replaces
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/softplus
for n in logit_n10k; do
  for mode in fused unfused; do
    if [ $mode = unfused ]; then export SWEEP_UNFUSE=1; else unset SWEEP_UNFUSE; fi
    echo "== $n $mode"
    python tools/sweep_user.py $n 8192 20 0x0 64x256 16x256 1x256
  done
done
unset SWEEP_UNFUSE
echo "== logit_bern_n10k"; python tools/sweep_user.py logit_bern_n10k 8192 20 0x0 64x256
for n in logistic_softplus records_logistic; do
  for mode in fused unfused; do
    if [ $mode = unfused ]; then export SWEEP_UNFUSE=1; else unset SWEEP_UNFUSE; fi
    echo "== $n $mode"; python tools/sweep_user.py $n 65536 200 0x0
  done
done
