#!/bin/bash
# A/B of the 16-slice hand-off schedule constants (MP_EXP = 1PPRRXX builds of tools/debug/build_exp.sh): headline + velocity phase counters
cd $GRAFT_REPO_ROOT
LIBS=""
for n in "$@"; do LIBS="$LIBS libmp_exp$n.so"; done
STEPS=${STEPS:-100} bash tools/debug/ab_libs.sh $LIBS
for n in "$@"; do
  echo "##### $n velocity layer 0 / 1"
  for l in 0 1; do MP_LIB_PATH=$PWD/mobileposer_amd/libmp_exp$n.so python tools/debug/prof_forward.py 3 $l 2>&1 | grep -v amdgpu | grep "x-proj\|validate\|total\|slow"; done
done
