#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r05_f_suite.txt
tail -6 gpurun_out/r05_f_suite.txt | cut -c1-250
for v in "" "wf=0"; do
MP_VARIANT=$v timeout 300 python bench.py --no-cpu-baseline --steps 200 2>gpurun_out/r05_f_bench_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant [$v]', d['ms_per_step'], {k[:34]:v['avg_launch_ms'] for k,v in d['kernels'].items() if isinstance(v,dict)})"
done
timeout 300 python tools/debug/prof_forward.py 3 0 256 wf 2>&1 | tail -4
