#!/bin/bash
# stability run of the round-6 binary (after the split of mp_api.hip and the per-sequence clusters): suite x 3 (the third under
# MP_GRAPH=2), fuzz 300 shapes + 400 small shapes against the plainest configuration, 10-minute soak
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_validation.txt
md5sum mobileposer_amd/libmobileposer_hip.so > $O
python -c "
import sys; sys.path.insert(0, '.')
from mobileposer_amd import _lib; print('build id', _lib.file_build_id())" >> $O
echo "== suite x 2 (eager)" >> $O
for i in 1 2; do timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu | grep -E "^E  |^FAILED|passed|failed" | head -40 >> $O; done
echo "== suite under MP_GRAPH=2" >> $O
MP_GRAPH=2 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu | grep -E "^E  |^FAILED|passed|failed" | head -40 >> $O
echo "== fuzz_shapes 300 (default vs plainest configuration)" >> $O
timeout 1500 python tools/debug/fuzz_shapes.py 300 2>&1 | grep -v amdgpu | tail -3 >> $O
echo "== fuzz_shapes 400 small (B <= 4)" >> $O
timeout 900 python tools/debug/fuzz_shapes.py 400 small 2>&1 | grep -v amdgpu | tail -3 >> $O
echo "== soak 600 s, exact fp32" >> $O
timeout 900 python tools/debug/soak.py 600 2>&1 | grep -v amdgpu | tail -3 >> $O
cat $O
