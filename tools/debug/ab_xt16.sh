#!/bin/bash
cd $GRAFT_REPO_ROOT
STEPS=100 bash tools/debug/ab_libs.sh libmobileposer_hip.so "$@"
for lib in libmobileposer_hip.so "$@"; do
  echo "##### $lib velocity layer 0 / 1"
  for l in 0 1; do MP_LIB_PATH=$PWD/mobileposer_amd/$lib python tools/debug/prof_forward.py 3 $l 2>&1 | grep -v amdgpu | grep "x-proj\|validate\|total\|slow"; done
done
