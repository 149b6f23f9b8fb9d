#!/bin/bash
# the whole suite on the wavefront build + timeline + phase counters of the two new launches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r05_e_suite.txt
tail -6 gpurun_out/r05_e_suite.txt
(cd /tmp && export TMPDIR=/tmp && timeout 600 python $GRAFT_REPO_ROOT/tools/debug/timeline.py 256 125) > gpurun_out/r05_e_timeline.txt 2>&1
tail -25 gpurun_out/r05_e_timeline.txt
timeout 300 python tools/debug/prof_forward.py 1 0 > gpurun_out/r05_e_prof_pose0.txt 2>&1; cat gpurun_out/r05_e_prof_pose0.txt | tail -9
timeout 300 python tools/debug/prof_forward.py 3 0 256 wf > gpurun_out/r05_e_prof_wf.txt 2>&1; cat gpurun_out/r05_e_prof_wf.txt | tail -12
timeout 300 python tools/debug/prof_forward.py 0 0 > gpurun_out/r05_e_prof_joints0.txt 2>&1; cat gpurun_out/r05_e_prof_joints0.txt | tail -9
