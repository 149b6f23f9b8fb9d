#!/bin/bash
mkdir -p gpurun_out
{
for v in "" "wf=0"; do
  MP_VARIANT="$v" timeout 300 python tools/debug/prof_replay.py 1000 2>&1 | grep -v amdgpu.ids
  MP_VARIANT="$v" timeout 300 python tools/debug/prof_b1.py 3 0 3000 2>&1 | grep -v amdgpu.ids
done
MP_VARIANT="wf=0" timeout 600 python tools/debug/online_timing.py 3000 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r05_v1_phases2.txt 2>&1
cat gpurun_out/r05_v1_phases2.txt
