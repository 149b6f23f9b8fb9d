#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do for lib in libmobileposer_hip.so libmp_xs3.so libmp_xs2.so libmp_xs5.so; do
  MP_LIB_PATH=$PWD/mobileposer_amd/$lib timeout 300 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-24s' % '$lib', d['ms_per_step'], d['output_sha1'][:8], {k[:24]:v['ms_per_forward'] for k,v in d['kernels'].items() if isinstance(v,dict)})"
done; done
