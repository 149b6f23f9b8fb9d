#!/bin/bash
# tagged-word hand-off (product) against the flagged hand-off of rounds 2-4 (libmp_exp0flags.so): bitwise outputs, then speed
cd $GRAFT_REPO_ROOT
for prof in init trained; do
  python tools/debug/sha_shapes.py $prof 2>&1 | grep -v amdgpu > gpurun_out/sha_tagx_$prof.txt
  MP_LIB_PATH=$PWD/mobileposer_amd/libmp_exp0flags.so python tools/debug/sha_shapes.py $prof 2>&1 | grep -v amdgpu > gpurun_out/sha_flags_$prof.txt
  if cmp -s gpurun_out/sha_tagx_$prof.txt gpurun_out/sha_flags_$prof.txt; then echo "$prof: IDENTICAL ($(wc -l < gpurun_out/sha_tagx_$prof.txt) lines)"; else echo "$prof: DIFFERENT"; diff gpurun_out/sha_tagx_$prof.txt gpurun_out/sha_flags_$prof.txt | head -20; fi
done
tail -3 gpurun_out/sha_tagx_init.txt
STEPS=100 bash tools/debug/ab_libs.sh libmobileposer_hip.so libmp_exp0flags.so "$@"
for l in 0 1; do python tools/debug/prof_forward.py 3 $l 2>&1 | grep -v amdgpu; done
python tools/debug/prof_forward.py 1 0 2>&1 | grep -v amdgpu
python tools/debug/prof_forward.py 1 1 2>&1 | grep -v amdgpu
