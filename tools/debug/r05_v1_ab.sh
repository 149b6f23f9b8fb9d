#!/bin/bash
# mp_lstm_v1: first poll before (default) / behind (MP_V1_EARLY_POLL=0 build, libmp_latepoll.so) the input projection, one box
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_errors.py -q -m gpu -x -k "single_sequence or replay or one_frame" 2>&1 | tail -4
for rep in 1 2; do
for lib in "" "$PWD/mobileposer_amd/libmp_latepoll.so"; do
  echo "== MP_LIB_PATH='$lib'"
  MP_LIB_PATH="$lib" timeout 600 python tools/debug/online_timing.py 3000 2>&1 | grep -v "amdgpu.ids\|replay classes"
done
done
timeout 300 python tools/debug/prof_b1.py 0 0 3000 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/debug/prof_replay.py 1000 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r05_v1_ab.txt 2>&1
cat gpurun_out/r05_v1_ab.txt
