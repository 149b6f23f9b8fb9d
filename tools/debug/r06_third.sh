#!/bin/bash
# round 6, third GPU call: the default bench line (timed), G17 both variants, the round-6 tests, tick latency at cadence
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
t0=$(date +%s)
MP_BENCH_TRACE_EVERY=120 timeout -s TERM 500 python bench.py > gpurun_out/r06_bench_first.json 2> gpurun_out/r06_bench_first.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"
python -c "
import json; d=json.load(open('gpurun_out/r06_bench_first.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['verified']['max_err']); print(d.get('configs0_single_sequence')); print(d.get('configs1_joints_only')); print(d.get('configs3_strong')); print(d.get('configs4_stream')); print(d['per_rank']); print(d['modes']); print(d['cpu_baseline'])" || grep -v amdgpu.ids gpurun_out/r06_bench_first.err | tail -40
timeout 1500 python -m pytest tests/test_gpu_round6.py -q -s 2>&1 | grep -v amdgpu.ids | tail -80 > gpurun_out/r06_round6_tests.txt; tail -70 gpurun_out/r06_round6_tests.txt
for S in 1 64 512; do for hz in 30 60; do
  timeout 300 python bench.py --workload stream --streams $S --cadence-hz $hz --steps 240 --warmup 20 2>gpurun_out/r06_cadence.err | tail -1 > gpurun_out/r06_cadence_S${S}_${hz}hz.json
  python -c "
import json; d=json.load(open('gpurun_out/r06_cadence_S${S}_${hz}hz.json'))
for k,v in d['modes'].items(): print('S=$S ${hz}Hz', k, 'b2b', v['back_to_back_ms'], 'cadence', v['at_cadence_ms'], 'misses', v['deadline_misses']); print('    1s idle:', v['after_1s_idle']['clock_before_mhz'], v['after_1s_idle']['tick_ms'][:8], v['after_1s_idle']['clock_after_tick']); print('    50ms idle:', v['after_50ms_idle']['clock_before_mhz'], v['after_50ms_idle']['tick_ms'][:6], v['after_50ms_idle']['clock_after_tick'])" || tail -5 gpurun_out/r06_cadence.err
done; done 2>&1 | tee gpurun_out/r06_tick_cadence_first.txt
