#!/bin/bash
# (GPU box) A/B of the fused softplus and of the 512-thread workgroups of closures with large staged data:
#   final   what the translator emits now             noopen  log1p_exp_v8 with its two branches per term
#   bt256   final, workgroups of at most 256 threads  unfused the two calls log1p_v8(exp_v8(eta)) (round 3), 256 threads
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { echo "== $1 [$2]"; AMWG_TRANSLATE_OPTS="$3" python tools/sweep_user.py $1 $4 $5 0x0 2>&1 | grep -v "^W2\|^E2"; }
for n in logit_n10k; do
  run $n final '{}' 8192 20
  run $n noopen '{"no_open_softplus":true}' 8192 20
  run $n bt256 '{"max_threads":256}' 8192 20
  SWEEP_UNFUSE=1 run $n unfused '{"no_open_softplus":true,"max_threads":256}' 8192 20
  run $n final '{}' 65536 10
done
run logit_bern_n10k final '{}' 8192 20
run logit_bern_n10k bt256 '{"max_threads":256}' 8192 20
run pois_const_rate final '{}' 65536 20
run pois_const_rate final '{}' 131072 20
