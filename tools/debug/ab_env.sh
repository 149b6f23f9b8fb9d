#!/bin/bash
# A/B of an environment switch on ONE box, alternating:  ab_env.sh VAR A B
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for v in $2 $3; do
  env $1=$v timeout 300 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1=$v', d['ms_per_step'], d['configs3_strong']['ms_per_step'])"
done; done
