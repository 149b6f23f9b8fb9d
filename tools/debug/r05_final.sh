#!/bin/bash
# round-5 evidence of ONE binary: suite, bench line (with CPU baseline), rocprofv3 stats + PMC, timeline, class times, configs, stream bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
md5sum mobileposer_amd/libmobileposer_hip.so > gpurun_out/r05_lib.md5
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/r05_suite.txt; tail -3 gpurun_out/r05_suite.txt
timeout 1500 python tools/profile.py r05 > gpurun_out/r05_profile.log 2>&1; tail -30 gpurun_out/r05_profile.log
cp gpurun_out/r05_pmc_summary.json profiles/r05_pmc_summary.json     # (bench.py quotes roofline.traffic from the summary of THIS binary)
timeout 600 python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r05_bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['end_to_end'], d['verified']['max_err'])"
(cd /tmp && export TMPDIR=/tmp && timeout 600 python $GRAFT_REPO_ROOT/tools/debug/timeline.py 256 125) > gpurun_out/r05_timeline_256x125.txt 2>&1; tail -14 gpurun_out/r05_timeline_256x125.txt
timeout 600 python tools/debug/class_times.py 128 256 1024 > gpurun_out/r05_class_times.txt 2>&1; cat gpurun_out/r05_class_times.txt
MP_VARIANT=wf=0 timeout 600 python tools/debug/class_times.py 256 1024 > gpurun_out/r05_class_times_wf0.txt 2>&1; cat gpurun_out/r05_class_times_wf0.txt
timeout 600 python tools/configs.py > gpurun_out/r05_configs.txt 2>&1; tail -12 gpurun_out/r05_configs.txt
timeout 300 python bench.py --workload stream --steps 100 --warmup 10 > gpurun_out/r05_bench_stream_fp32.json 2>/dev/null; cat gpurun_out/r05_bench_stream_fp32.json | cut -c1-400
timeout 600 python tools/debug/online_timing.py 3000 2>&1 | tail -4 > gpurun_out/r05_online_timing.txt; cat gpurun_out/r05_online_timing.txt
timeout 300 bash tools/debug/pmc_forward.sh > gpurun_out/r05_pmc_forward.txt 2>&1; cat gpurun_out/r05_pmc_forward.txt
# the one-sequence kernels (mp_lstm_v1 / v1s) against the MFMA path at B = 1: timings, timeline of one 3000-frame sequence, phase counters
{ for v in "" "vec=0"; do echo "== MP_VARIANT='$v'"; MP_VARIANT="$v" timeout 600 python tools/debug/online_timing.py 3000 2>&1 | grep -v amdgpu.ids; done; } > gpurun_out/r05_v1_timing.txt 2>&1; cat gpurun_out/r05_v1_timing.txt
(cd /tmp && export TMPDIR=/tmp && timeout 600 python $GRAFT_REPO_ROOT/tools/debug/timeline.py 1 3000) > gpurun_out/r05_timeline_1x3000.txt 2>&1; tail -22 gpurun_out/r05_timeline_1x3000.txt
{ for v in "" "vec=0"; do for ml in "0 0" "0 1"; do MP_VARIANT="$v" timeout 300 python tools/debug/prof_b1.py $ml 3000 2>&1 | grep -v amdgpu.ids; done; done; timeout 300 python tools/debug/prof_replay.py 1000 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r05_v1_phases.txt 2>&1; cat gpurun_out/r05_v1_phases.txt
