#!/bin/bash
# round 6, sixth GPU call: the whole GPU suite on the split sources, the bench sha, the rocprofv3 profile of bench.py
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/r06_suite.txt; tail -12 gpurun_out/r06_suite.txt
timeout 300 python bench.py > gpurun_out/r06_bench.json 2>gpurun_out/r06_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r06_bench.json')); print('bench', d['ms_per_step'], d['value'], d['output_sha1'], d['build_id'], d['roofline']['frac'], d['roofline']['traffic'])"
timeout 1500 python tools/profile.py r06 2>&1 | grep -v amdgpu.ids | tail -60
head -40 gpurun_out/r06_kernel_stats.md
