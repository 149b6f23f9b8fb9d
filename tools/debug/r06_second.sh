#!/bin/bash
# round 6, second GPU call: where does bench.py hang, why does the cadence mode die, the rest of the round-6 tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s TERM 300 python bench.py --steps 20 --no-cpu-baseline > gpurun_out/r06_bench_dbg.json 2> gpurun_out/r06_bench_dbg.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r06_bench_dbg.json; grep -v amdgpu.ids gpurun_out/r06_bench_dbg.err | tail -40
timeout -s TERM 200 python bench.py --workload stream --streams 64 --cadence-hz 60 --steps 60 --warmup 10 > gpurun_out/r06_cad_dbg.json 2> gpurun_out/r06_cad_dbg.err; echo "cadence rc=$?"; tail -c 400 gpurun_out/r06_cad_dbg.json; grep -v amdgpu.ids gpurun_out/r06_cad_dbg.err | tail -30
timeout 1500 python -m pytest tests/test_gpu_round6.py -q -s --deselect "tests/test_gpu_round6.py::test_g17_single_sequence_trained_regime" 2>&1 | grep -v amdgpu.ids | tail -60 > gpurun_out/r06_round6_tests2.txt; tail -50 gpurun_out/r06_round6_tests2.txt
