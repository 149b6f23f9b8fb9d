#!/bin/bash
# phase counters of the joints layers for the product library and the MP_EXP variants given as arguments
cd $GRAFT_REPO_ROOT
for n in 0 "$@"; do
  lib=libmp_exp$n.so; [ $n = 0 ] && lib=libmobileposer_hip.so
  echo "##### MP_EXP=$n"
  MP_LIB_PATH=$PWD/mobileposer_amd/$lib MP_PERSIST_PROF=1 timeout 300 python tools_prof_persist.py ${MOD:-joints} 2>&1 | grep -v amdgpu
done
