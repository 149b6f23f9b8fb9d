#!/bin/bash
# reproduce the mp_stream_step SIGSEGV (GPUTEST_r01) and narrow it down
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dbg
export PYTHONFAULTHANDLER=1
echo "=== A: isolated test, default"; timeout 300 python -m pytest tests/test_gpu_parity.py -k multi_stream -x -q -p no:cacheprovider > gpurun_out/dbg/A.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/dbg/A.log
echo "=== B: isolated test, MP_NO_GRAPH=1"; MP_NO_GRAPH=1 timeout 300 python -m pytest tests/test_gpu_parity.py -k multi_stream -x -q -p no:cacheprovider > gpurun_out/dbg/B.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/dbg/B.log
echo "=== C: full suite"; timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/dbg/C.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/dbg/C.log | cut -c1-300
echo "=== D: gdb on isolated"; timeout 600 rocgdb -batch -ex "handle SIGSEGV stop print" -ex run -ex bt -ex "info threads" --args python -m pytest tests/test_gpu_parity.py -k multi_stream -x -q -p no:cacheprovider > gpurun_out/dbg/D.log 2>&1; echo "rc=$?"; grep -n -A40 "SIGSEGV" gpurun_out/dbg/D.log | head -80
echo "=== E: full suite under gdb"; timeout 900 rocgdb -batch -ex "handle SIGSEGV stop print" -ex run -ex bt --args python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/dbg/E.log 2>&1; echo "rc=$?"; grep -n -A40 "SIGSEGV" gpurun_out/dbg/E.log | head -80; tail -5 gpurun_out/dbg/E.log | cut -c1-300
