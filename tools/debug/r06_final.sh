#!/bin/bash
# round-6 evidence of ONE binary: suite, bench line (with CPU baseline), rocprofv3 stats + PMC (headline legs) and the all-legs
# summary, timeline, class times, the other configs, ticks back to back and at a cadence, small batches, one-sequence timings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
md5sum mobileposer_amd/libmobileposer_hip.so > gpurun_out/r06_lib.md5
timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_suite_full.txt; grep -E "^E  |^FAILED|passed|failed|200 lengths|first call|G17|tick: clean|vs mp_lstm_u8|64 x 60 forward" gpurun_out/r06_suite_full.txt | cut -c1-260 > gpurun_out/r06_suite.txt; tail -12 gpurun_out/r06_suite.txt
timeout 1800 python tools/profile.py r06 > gpurun_out/r06_profile.log 2>&1; tail -5 gpurun_out/r06_profile.log
cp gpurun_out/r06_pmc_summary.json profiles/r06_pmc_summary.json     # (bench.py quotes roofline.traffic from the summary of THIS binary)
timeout 600 python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r06_bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['end_to_end'], d['verified']['max_err'], d['output_sha1'][:12], d['build_id'][:12])"
(cd /tmp && export TMPDIR=/tmp && timeout 600 python $GRAFT_REPO_ROOT/tools/debug/timeline.py 256 125) > gpurun_out/r06_timeline_256x125.txt 2>&1; tail -14 gpurun_out/r06_timeline_256x125.txt
timeout 600 python tools/debug/class_times.py 128 256 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_class_times.txt; cat gpurun_out/r06_class_times.txt
timeout 600 python tools/configs.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_configs.txt; tail -12 gpurun_out/r06_configs.txt
timeout 300 python bench.py --workload stream --steps 100 --warmup 10 > gpurun_out/r06_bench_stream_fp32.json 2>/dev/null; cut -c1-300 gpurun_out/r06_bench_stream_fp32.json
for S in 1 64 512; do for hz in 30 60; do
  timeout 300 python bench.py --workload stream --streams $S --cadence-hz $hz --steps 240 --warmup 20 2>/dev/null | tail -1 > gpurun_out/r06_cadence_S${S}_${hz}hz.json
  python -c "
import json; d=json.load(open('gpurun_out/r06_cadence_S${S}_${hz}hz.json'))
for k,v in d['modes'].items(): print('S=$S ${hz}Hz', k, 'b2b', v['back_to_back_ms'], 'cadence', v['at_cadence_ms'], 'misses', v['deadline_misses']); print('    1s idle:', v['after_1s_idle']['clock_before_mhz'], v['after_1s_idle']['tick_ms'][:8], v['after_1s_idle']['clock_after_tick']); print('    50ms idle:', v['after_50ms_idle']['clock_before_mhz'], v['after_50ms_idle']['tick_ms'][:6], v['after_50ms_idle']['clock_after_tick'])"
done; done > gpurun_out/r06_tick_cadence.txt 2>&1; grep cadence gpurun_out/r06_tick_cadence.txt | cut -c1-200
timeout 600 python tools/debug/small_batches.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_small_batches.txt; head -24 gpurun_out/r06_small_batches.txt
timeout 600 python tools/debug/online_timing.py 3000 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r06_online_timing.txt; cat gpurun_out/r06_online_timing.txt
timeout 300 python tools/pcie_inclusive.py 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/r06_pcie_inclusive.json; cut -c1-400 gpurun_out/r06_pcie_inclusive.json
