#!/bin/bash
# stability run of the round-5 binary: suite x 3, fuzz 300 shapes, 15-minute soak
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_validation.txt
md5sum mobileposer_amd/libmobileposer_hip.so > $O
python -c "
import sys; sys.path.insert(0, '.')
from mobileposer_amd import _lib; print('build id', _lib.file_build_id())" >> $O
echo "== suite x 3" >> $O
bash tools/debug/suite.sh 3 >> $O 2>&1
echo "== fuzz_shapes 300 (default vs plainest configuration)" >> $O
python tools/debug/fuzz_shapes.py 300 2>&1 | grep -v amdgpu | tail -3 >> $O
echo "== soak 900 s, exact fp32" >> $O
python tools/debug/soak.py 900 2>&1 | grep -v amdgpu | tail -3 >> $O
cat $O
