#!/bin/bash
# development aid: builds libamwg variants with one piece of the group-local step cut out (AMWG_X_GLCUT = n, wrong results) into build/cut/,
# for pricing the pieces on the GPU box:   AMWG_LIB=build/cut/libamwg_cut3.so python bench.py --workload cfg4 --group-local ...
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/bayes.js_amd/csrc
mkdir -p $R/build/cut
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -falign-loops=64 -Wno-unused-function"
build_one() {
  n=$1
  cd $C && /opt/rocm/bin/hipcc $FLAGS -DAMWG_FAMILY=2 -DAMWG_X_GLCUT=$n $2 -c -o $R/build/cut/k2_$n.o amwg_kernels.hip 2>/dev/null &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/cut/libamwg_cut$n.so amwg_core.o amwg_kernels_0.o amwg_kernels_1.o $R/build/cut/k2_$n.o amwg_kernels_3.o amwg_summaries.o amwg_group.o amwg_rtc_headers.o -lhiprtc -ldl && echo built $n
}
for n in "$@"; do build_one $n & done
wait
