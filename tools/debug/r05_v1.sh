#!/bin/bash
# the single-sequence kernels (mp_lstm_v1 / mp_lstm_v1s) against the MFMA kernels: tests, then the B = 1 paths timed
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_round5.py -q -m gpu -x -k "single_sequence or replay" -s 2>&1 | tail -15
for v in "" "vec=0"; do
  echo "== MP_VARIANT='$v'"
  MP_VARIANT="$v" timeout 600 python tools/debug/online_timing.py 3000 2>&1 | grep -v amdgpu.ids
done
(cd /tmp && export TMPDIR=/tmp && python $GRAFT_REPO_ROOT/tools/debug/timeline.py 1 3000 2>&1 | tail -25)
(cd /tmp && export TMPDIR=/tmp && python $GRAFT_REPO_ROOT/tools/debug/timeline.py 1 45 2>&1 | tail -25)
} > gpurun_out/r05_v1.txt 2>&1
tail -75 gpurun_out/r05_v1.txt
