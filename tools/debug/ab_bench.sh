#!/bin/bash
# A/B on ONE box: mobileposer_amd/libmp_old.so against the current library, alternating, fp32 headline
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for lib in libmp_old.so libmobileposer_hip.so; do
    MP_LIB_PATH=$PWD/mobileposer_amd/$lib timeout 300 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['ms_per_step'], d['roofline']['avg_launch_ms'], {k[:28]:v['avg_launch_ms'] for k,v in d['kernels'].items() if isinstance(v,dict)})"
  done
done
