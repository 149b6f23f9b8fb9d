#!/bin/bash
# round 6, eighth GPU call: groups of four sequences on mp_lstm_v1 (5 <= B <= 16): tests, timing
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py -q -x -k "few_sequences or single_sequence_kernel or one_frame_calls or starved or glitching" 2>&1 | grep -v amdgpu.ids | tail -30 > gpurun_out/r06_seq_tests.txt; tail -25 gpurun_out/r06_seq_tests.txt
timeout 600 python tools/debug/small_batches.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_small_batches.txt
timeout 600 python tools/debug/fuzz_shapes.py 100 small 2>&1 | grep -v amdgpu.ids | tail -3
