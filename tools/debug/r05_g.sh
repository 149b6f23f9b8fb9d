#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do for v in "" "tail3=0"; do
MP_VARIANT=$v timeout 300 python bench.py --no-cpu-baseline --steps 200 2>gpurun_out/r05_g_bench_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant [$v]', d['ms_per_step'], d['config']['recoveries_during_run'], d['kernels']['forward_event_timed_ms'], {k[:24]:(v['launches_per_forward'], v['ms_per_forward']) for k,v in d['kernels'].items() if isinstance(v,dict)})"
done; done
