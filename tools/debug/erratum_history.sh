#!/bin/bash
# Reproducer for profiles/NOTES_r01-r03.md 4.3 (gfx950: packed-fp32 VALU results of one kernel wrong in lanes 48..63 while certain
# revisions of the split-bf16 LSTM kernel run on the same CUs).  Step 1 (any machine with hipcc): rebuild historical
# revisions of this repository WITH packed-fp32 instructions enabled:   tools/debug/erratum_history.sh build
# Step 2 (MI355X):                                                       tools/debug/erratum_history.sh run
# Every revision contains the byte-identical mp_r6d_ik ISA (525 instructions, md5 6ae61ad1 of the mnemonic text); the
# probe compares the pose that mp_forward's IK kernel wrote (it runs on a side stream beside the velocity layers)
# with the same kernel re-run alone on the same input.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
REVS="5908408 cfce2ad bc895bc 25fbc48 11a22c5 db6eba1 5981f33 83df14a 6e90bf4"
if [ "$1" = build ]; then
  for sha in $REVS; do
    d=$ROOT/tools/debug/old_$sha; rm -rf $d; mkdir -p $d
    git -C $ROOT archive $sha | tar -x -C $d
    sed -i 's/"-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"\]/]/' $d/__graft_entry__.py
    (cd $d && python __graft_entry__.py --force | tail -1)
    sed "s/REV/$sha/" $ROOT/tools/debug/probe_old.py > $d/probe_old.py
  done
else
  for sha in $REVS; do (cd $ROOT/tools/debug/old_$sha && timeout 300 python probe_old.py 3 30 2>&1 | grep -v amdgpu.ids | tail -3); done
fi
