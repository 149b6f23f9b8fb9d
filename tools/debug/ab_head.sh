#!/bin/bash
# product library against libmp_exp0head.so (mp_lstm_persist.hip of the last commit): bitwise outputs, then speed, then phase counters
cd $GRAFT_REPO_ROOT
for prof in init trained; do
  python tools/debug/sha_shapes.py $prof 2>&1 | grep -v amdgpu > gpurun_out/sha_new_$prof.txt
  MP_LIB_PATH=$PWD/mobileposer_amd/libmp_exp0head.so python tools/debug/sha_shapes.py $prof 2>&1 | grep -v amdgpu > gpurun_out/sha_head_$prof.txt
  if cmp -s gpurun_out/sha_new_$prof.txt gpurun_out/sha_head_$prof.txt; then echo "$prof: IDENTICAL ($(wc -l < gpurun_out/sha_new_$prof.txt) lines)"; else echo "$prof: DIFFERENT"; diff gpurun_out/sha_new_$prof.txt gpurun_out/sha_head_$prof.txt | head -20; fi
done
STEPS=100 bash tools/debug/ab_libs.sh libmobileposer_hip.so libmp_exp0head.so
for ml in "1 0" "1 1" "3 0"; do python tools/debug/prof_forward.py $ml 2>&1 | grep -v amdgpu; done
