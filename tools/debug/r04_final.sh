#!/bin/bash
# last run of the round: refresh every binary-dependent profile, then the bench line with that PMC profile, then validation
cd $GRAFT_REPO_ROOT
bash tools/debug/r04_refresh.sh > gpurun_out/r04_refresh.log 2>&1
cp gpurun_out/r04_pmc_summary.json profiles/r04_pmc_summary.json      # (bench.py quotes traffic only from a profile of THIS binary)
O=gpurun_out/r04_validation4.txt
python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err
echo "== suite x 3" > $O
bash tools/debug/suite.sh 3 >> $O 2>&1
echo "== suite under MP_GRAPH=2" >> $O
MP_GRAPH=2 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3 >> $O
echo "== suite with MP_LSTM_MODE=x3 as the handle default" >> $O
MP_LSTM_MODE=x3 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3 >> $O
echo "== fuzz_modes 150" >> $O
python tools/debug/fuzz_modes.py 150 2>&1 | grep -v amdgpu | tail -3 >> $O
echo "== fuzz_shapes 120" >> $O
python tools/debug/fuzz_shapes.py 120 2>&1 | grep -v amdgpu | tail -2 >> $O
echo "== soak 240 s, exact fp32" >> $O
python tools/debug/soak.py 240 2>&1 | grep -v amdgpu | tail -3 >> $O
cat $O
python -c "
import json; d=json.loads(open('gpurun_out/r04_bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['end_to_end'], d['modes']['x3'], d['configs3_strong']['ms_per_step'])"
cat gpurun_out/r04_configs.txt gpurun_out/r04_class_times.txt
