#!/bin/bash
# the whole -m gpu suite (no -x: every failure is listed) + a bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/r05_b_suite.txt
timeout 400 python bench.py --steps 100 --no-cpu-baseline > gpurun_out/r05_b_bench.json 2> gpurun_out/r05_b_bench.err
tail -15 gpurun_out/r05_b_suite.txt; python -c "
import json; d=json.load(open('gpurun_out/r05_b_bench.json')); print(d['ms_per_step'], d['verified']['max_err'], {k[:30]:v['avg_launch_ms'] for k,v in d['kernels'].items() if isinstance(v,dict)})"
