#!/bin/bash
mkdir -p gpurun_out
{
for v in "" "vec=0"; do
  for ml in "0 0" "0 1" "3 0"; do
    MP_VARIANT="$v" timeout 300 python tools/debug/prof_b1.py $ml 3000 2>&1 | grep -v amdgpu.ids
  done
done
} > gpurun_out/r05_v1_phases.txt 2>&1
cat gpurun_out/r05_v1_phases.txt
