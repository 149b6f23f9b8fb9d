#!/bin/bash
# the one-stream tick: snapshot copies as one launch, deep-prefetch linear layers for a handful of rows (MP_VARIANT gemm_few=0: off)
mkdir -p gpurun_out
{
timeout 1200 python -m pytest tests -q -m gpu -x -k "online or stream or tick or replay or single_sequence or golden or ragged or permut or determin or graph" 2>&1 | tail -4
for v in "" "gemm_few=0"; do
  echo "== MP_VARIANT='$v'"
  MP_VARIANT="$v" timeout 600 python tools/debug/online_timing.py 3000 2>&1 | grep -v "amdgpu.ids\|replay classes"
done
(cd /tmp && export TMPDIR=/tmp && python $GRAFT_REPO_ROOT/tools/debug/timeline.py 0 45 2>&1 | tail -22)
} > gpurun_out/r05_tick.txt 2>&1
cat gpurun_out/r05_tick.txt
