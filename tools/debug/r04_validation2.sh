#!/bin/bash
# second long validation (different box): suite x 8, erratum soak, accuracy for three more weight seeds at 64 x 125
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_validation2.txt
echo "== suite x 8" > $O
bash tools/debug/suite.sh 8 >> $O 2>&1
echo "== suite with MP_LSTM_MODE=x3 as the handle default" >> $O
MP_LSTM_MODE=x3 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3 >> $O
echo "== fuzz_shapes (default configuration vs round-1 configuration), 120 cases" >> $O
python tools/debug/fuzz_shapes.py 120 2>&1 | grep -v amdgpu | tail -2 >> $O
cat $O
