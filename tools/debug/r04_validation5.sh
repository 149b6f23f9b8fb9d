#!/bin/bash
# long stability run of the final binary (~55 min): suite x 6, erratum check, 15-minute soak per operand mode
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_validation5.txt
echo "== suite x 6" > $O
bash tools/debug/suite.sh 6 >> $O 2>&1
echo "== soak 900 s, exact fp32" >> $O
python tools/debug/soak.py 900 2>&1 | grep -v amdgpu | tail -3 >> $O
echo "== soak 900 s, split-fp16" >> $O
MP_LSTM_MODE=x3 python tools/debug/soak.py 900 2>&1 | grep -v amdgpu | tail -3 >> $O
echo "== fuzz_modes 300" >> $O
python tools/debug/fuzz_modes.py 300 2>&1 | grep -v amdgpu | tail -3 >> $O
cat $O
