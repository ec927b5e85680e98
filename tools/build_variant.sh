#!/bin/bash
# Development aid: an A/B build of libamwg.so with extra compiler flags, next to the product build.
#   tools/build_variant.sh <name> [-DAMWG_STEPPER_PRIORITY=0 ...]   ->  build/<name>/libamwg.so   (use with AMWG_LIB=build/<name>/libamwg.so)
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/build/$NAME
mkdir -p $D/bayes.js_amd $D/include
mkdir -p $D/bayes.js_amd/csrc
for f in $R/bayes.js_amd/csrc/*.h $R/bayes.js_amd/csrc/*.hip $R/bayes.js_amd/csrc/*.c $R/bayes.js_amd/csrc/Makefile; do
  cmp -s $f $D/bayes.js_amd/csrc/$(basename $f) || cp $f $D/bayes.js_amd/csrc/
done
cp $R/include/*.h $D/include/
mkdir -p $D/tools && cp $R/tools/build_id.py $D/tools/
make -s -j8 -C $D/bayes.js_amd/csrc libamwg.so HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -falign-loops=64 -Wall -Wno-unused-function $*"
cp $D/bayes.js_amd/csrc/libamwg.so $D/libamwg.so
echo "built $D/libamwg.so"
