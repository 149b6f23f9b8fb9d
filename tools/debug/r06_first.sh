#!/bin/bash
# round 6, first GPU call: the new tests, the l2l1 A/B, the bench line with its new legs, tick latency at cadence, mode-3 accuracy
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_round6.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -60 > gpurun_out/r06_round6_tests.txt; tail -40 gpurun_out/r06_round6_tests.txt
timeout 900 python tools/debug/ab_variants.py 256 3 l2l1_opt=0 l2l1_opt=1 l2l1_opt=2 l2l1_opt=3 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_l2l1_ab.txt; cat gpurun_out/r06_l2l1_ab.txt
timeout 300 python tools/debug/class_times.py 128 256 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_class_times_first.txt; cat gpurun_out/r06_class_times_first.txt
timeout 600 python bench.py > gpurun_out/r06_bench_first.json 2> gpurun_out/r06_bench_first.err; python -c "
import json; d=json.load(open('gpurun_out/r06_bench_first.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['verified']['max_err']); print(d.get('configs0_single_sequence')); print(d.get('configs1_joints_only')); print(d.get('configs3_strong')); print(d.get('configs4_stream')); print(d['per_rank']); print(d['modes'])" || tail -20 gpurun_out/r06_bench_first.err
for S in 1 64 512; do for hz in 30 60; do
  timeout 300 python bench.py --workload stream --streams $S --cadence-hz $hz --steps 240 --warmup 20 2>/dev/null | tail -1 > gpurun_out/r06_cadence_S${S}_${hz}hz.json
  python -c "
import json; d=json.load(open('gpurun_out/r06_cadence_S${S}_${hz}hz.json'))
for k,v in d['modes'].items(): print('S=$S ${hz}Hz', k, 'b2b', v['back_to_back_ms'], 'cadence', v['at_cadence_ms'], 'misses', v['deadline_misses']); print('    1s idle:', v['after_1s_idle']['clock_before_mhz'], v['after_1s_idle']['tick_ms'][:8], v['after_1s_idle']['clock_after_tick']); print('    50ms idle:', v['after_50ms_idle']['clock_before_mhz'], v['after_50ms_idle']['tick_ms'][:6], v['after_50ms_idle']['clock_after_tick'])"
done; done 2>&1 | tee gpurun_out/r06_tick_cadence_first.txt
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q -s -k "baseline_size_vs_oracle_other_weights" 2>&1 | grep -v amdgpu.ids | grep -E "vs float64|vs fp32|passed|failed|Error" > gpurun_out/r06_mode3_first.txt; cat gpurun_out/r06_mode3_first.txt
