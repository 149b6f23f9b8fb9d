#!/bin/bash
# round 6, fifth GPU call: G17 on the 16-member band, the loaded clock probe after idle, suite, bench sha
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round6.py -q -s -k "g17 or mode3" 2>&1 | grep -v amdgpu.ids | tail -40 > gpurun_out/r06_g17_tests.txt; tail -32 gpurun_out/r06_g17_tests.txt
timeout 600 python tools/debug/idle_probe.py 512 0.016 1.0 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_idle_probe_S512.txt; cat gpurun_out/r06_idle_probe_S512.txt | cut -c1-600
timeout 300 python bench.py --steps 100 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['output_sha1'], d['build_id'], d['roofline']['frac'])"
