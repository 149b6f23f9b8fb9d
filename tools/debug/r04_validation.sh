#!/bin/bash
# Round-4 long validation on one GPU box (~45 min): suite x 6 (one of them under MP_GRAPH=2), fuzz of modes / schedules over
# 300 random shapes, 10-minute soak per operand mode, accuracy at 256 x 125 for three weight seeds and both profiles.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_validation.txt
echo "== suite x 5" > $O
bash tools/debug/suite.sh 5 >> $O 2>&1
echo "== suite under MP_GRAPH=2" >> $O
MP_GRAPH=2 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3 >> $O
echo "== fuzz 150 + 150" >> $O
python tools/debug/fuzz_modes.py 150 2>&1 | grep -v amdgpu | tail -3 >> $O
echo "== soak 600 s, exact fp32" >> $O
python tools/debug/soak.py 600 2>&1 | grep -v amdgpu | tail -3 >> $O
echo "== soak 600 s, split-fp16" >> $O
MP_LSTM_MODE=x3 python tools/debug/soak.py 600 2>&1 | grep -v amdgpu | tail -3 >> $O
echo "== accuracy 256 x 125" >> $O
python tools/accuracy.py 256 125 2>&1 | grep -v amdgpu >> $O
cat $O
