#!/bin/bash
# round 6, fourth GPU call: what an idle GPU costs (idle_probe), then the whole GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/debug/idle_probe.py 512 0.016 0.05 1.0 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_idle_probe_S512.txt; cat gpurun_out/r06_idle_probe_S512.txt
timeout 300 python tools/debug/idle_probe.py 1 0.016 1.0 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_idle_probe_S1.txt; cat gpurun_out/r06_idle_probe_S1.txt
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_round6.py::test_g17_single_sequence_trained_regime 2>&1 | grep -v amdgpu.ids | tail -40 > gpurun_out/r06_suite_first.txt; tail -30 gpurun_out/r06_suite_first.txt
