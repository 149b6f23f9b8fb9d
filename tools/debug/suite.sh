#!/bin/bash
# run the whole GPU suite N times on this box; logs of failing runs are kept
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite
n=${1:-1}
for i in $(seq 1 $n); do
  timeout 1200 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/suite/run_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(grep -E ' passed| failed| error' gpurun_out/suite/run_$i.log | tail -1)"
  if [ $rc -ne 0 ]; then grep -n -E "^(FAILED|ERROR)|Error|assert|Fatal|fault" gpurun_out/suite/run_$i.log | head -40; fi
done
