#!/bin/bash
# Refresh every profiles/r04_* number that depends on the library binary (run on the GPU box after the last kernel change).
cd $GRAFT_REPO_ROOT
G=gpurun_out
python tools/profile.py r04 > $G/r04_profile.log 2>&1
python bench.py > $G/r04_bench.json 2> $G/r04_bench.err
python bench.py --workload stream --lstm-mode fp32 --no-cpu-baseline > $G/r04_bench_stream_fp32.json 2>/dev/null
python bench.py --workload stream --lstm-mode x3 --no-cpu-baseline > $G/r04_bench_stream_x3.json 2>/dev/null
python tools/configs.py 2>&1 | grep -v amdgpu > $G/r04_configs.txt
python tools/debug/class_times.py 128 256 1024 2>&1 | grep -v amdgpu > $G/r04_class_times.txt
python tools/debug/timeline.py 256 125 2>&1 | grep -v amdgpu > $G/r04_timeline_256x125.txt
MP_TL_MODE=3 python tools/debug/timeline.py 256 125 2>&1 | grep -v amdgpu > $G/r04_timeline_256x125_mode3.txt
bash tools/debug/pmc_forward.sh > $G/r04_pmc_forward_raw.txt 2>&1
python tools/accuracy.py > $G/r04_accuracy.log 2>&1
MP_ACCURACY_OUT=r04_accuracy_256x125.json python tools/accuracy.py 256 125 > $G/r04_accuracy_256x125.log 2>&1
tail -2 $G/r04_bench.json; cat $G/r04_configs.txt $G/r04_class_times.txt; head -16 $G/r04_timeline_256x125.txt; ls $G | head -50
