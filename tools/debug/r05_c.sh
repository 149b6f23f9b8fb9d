#!/bin/bash
# wavefront first light: parity at the sizes that take the wavefront path, then a bench line and an A/B against wf=0
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest -x -q -m gpu "tests/test_gpu_parity.py::test_full_size_vs_oracle_and_properties" "tests/test_gpu_round5.py::test_the_call_bench_times_vs_oracle" tests/test_gpu_parity.py::test_fused_fp32_kernels_match_the_per_step_kernels_on_ragged_launch_groups tests/test_gpu_parity.py::test_hidden_state_transports_agree 2>&1 | tail -30 > gpurun_out/r05_c_tests.txt
tail -12 gpurun_out/r05_c_tests.txt
for v in "" "wf=0"; do
MP_VARIANT=$v timeout 300 python bench.py --no-cpu-baseline --steps 100 2>gpurun_out/r05_c_bench_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant [$v]', d['ms_per_step'], d['verified']['max_err'], {k[:34]:v['avg_launch_ms'] for k,v in d['kernels'].items() if isinstance(v,dict)})"
done
