#!/bin/bash
# round 5, first GPU call: the whole -m gpu suite on the round-5 host changes, a bench line, tanh accuracy A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r05_a_suite.txt
timeout 400 python bench.py --steps 100 > gpurun_out/r05_a_bench.json 2> gpurun_out/r05_a_bench.err
MP_ACCURACY_MODES=1 MP_ACCURACY_OUT=r05_accuracy_256x125_base.json timeout 900 python tools/accuracy.py 256 125 > gpurun_out/r05_a_acc_base.txt 2>&1
MP_LIB_PATH=$PWD/mobileposer_amd/libmp_tanhpoly.so MP_ACCURACY_MODES=1 MP_ACCURACY_OUT=r05_accuracy_256x125_tanhpoly.json timeout 900 python tools/accuracy.py 256 125 > gpurun_out/r05_a_acc_poly.txt 2>&1
for i in 1 2; do for lib in libmobileposer_hip.so libmp_tanhpoly.so; do
  MP_LIB_PATH=$PWD/mobileposer_amd/$lib timeout 300 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-24s' % '$lib', d['ms_per_step'], {k[:26]:v['avg_launch_ms'] for k,v in d['kernels'].items() if isinstance(v,dict)})"
done; done > gpurun_out/r05_a_ab.txt 2>&1
tail -5 gpurun_out/r05_a_suite.txt; cat gpurun_out/r05_a_ab.txt
