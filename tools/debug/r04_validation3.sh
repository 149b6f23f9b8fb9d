#!/bin/bash
# Validation of the tagged hand-off build (~35 min): bench line with PMC traffic, suite x 4 + once under MP_GRAPH=2 + once with
# mode 3 as the handle default, fuzz of modes / schedules, 5-minute soak per operand mode, stream benches.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_validation3.txt
python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err
python bench.py --workload stream --lstm-mode fp32 --no-cpu-baseline > gpurun_out/r04_bench_stream_fp32.json 2>/dev/null
python bench.py --workload stream --lstm-mode x3 --no-cpu-baseline > gpurun_out/r04_bench_stream_x3.json 2>/dev/null
echo "== suite x 4" > $O
bash tools/debug/suite.sh 4 >> $O 2>&1
echo "== suite under MP_GRAPH=2" >> $O
MP_GRAPH=2 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3 >> $O
echo "== suite with MP_LSTM_MODE=x3 as the handle default" >> $O
MP_LSTM_MODE=x3 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3 >> $O
echo "== fuzz_modes 150" >> $O
python tools/debug/fuzz_modes.py 150 2>&1 | grep -v amdgpu | tail -3 >> $O
echo "== fuzz_shapes 120" >> $O
python tools/debug/fuzz_shapes.py 120 2>&1 | grep -v amdgpu | tail -2 >> $O
echo "== soak 300 s, exact fp32" >> $O
python tools/debug/soak.py 300 2>&1 | grep -v amdgpu | tail -3 >> $O
echo "== soak 300 s, split-fp16" >> $O
MP_LSTM_MODE=x3 python tools/debug/soak.py 300 2>&1 | grep -v amdgpu | tail -3 >> $O
cat $O
python -c "
import json; d=json.loads(open('gpurun_out/r04_bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'], d['modes']['x3'])"
