#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dbg
export PYTHONFAULTHANDLER=1
T="python -m pytest tests/test_gpu_parity.py -k multi_stream -x -q -p no:cacheprovider"
echo "=== A: default"; timeout 300 $T > gpurun_out/dbg/A2.log 2>&1; echo "rc=$?"
echo "=== F: GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 timeout 300 $T > gpurun_out/dbg/F.log 2>&1; echo "rc=$?"
echo "=== G: GPU_MAX_HW_QUEUES=2"; GPU_MAX_HW_QUEUES=2 timeout 300 $T > gpurun_out/dbg/G.log 2>&1; echo "rc=$?"
echo "=== H: DEBUG_HIP_FORCE_GRAPH_QUEUES=1"; DEBUG_HIP_FORCE_GRAPH_QUEUES=1 timeout 300 $T > gpurun_out/dbg/H.log 2>&1; echo "rc=$?"
echo "=== bench graph"; timeout 600 python bench.py --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/dbg/bench_graph.json 2>gpurun_out/dbg/bench_graph.err; echo "rc=$?"
echo "=== bench nograph"; timeout 600 python bench.py --no-cpu-baseline --no-graph --steps 100 --warmup 10 > gpurun_out/dbg/bench_nograph.json 2>gpurun_out/dbg/bench_nograph.err; echo "rc=$?"
python - <<'PY'
import json
for n in ("graph","nograph"):
    try:
        d=json.loads(open("gpurun_out/dbg/bench_%s.json"%n).read().strip().splitlines()[-1])
        print(n, d["modes"])
    except Exception as e: print(n, "ERR", e)
PY
