#!/bin/bash
# split-fp16 mode: two builds of the library on one box, alternating
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for lib in "$@"; do
    MP_LIB_PATH=$PWD/mobileposer_amd/$lib timeout 300 python bench.py --no-cpu-baseline --lstm-mode x3 --steps 150 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-24s' % '$lib', d['ms_per_step'], {k[:22]:v['avg_launch_ms'] for k,v in d['kernels'].items() if isinstance(v,dict)})"
  done
done
