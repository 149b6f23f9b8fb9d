#!/bin/bash
# after the last (small) kernel change of the round: suite once, PMC / rocprof profile of this binary, bench line quoting it
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_validation6.txt
echo "== suite" > $O
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4 >> $O
python tools/profile.py r04 > gpurun_out/r04_profile.log 2>&1
cp gpurun_out/r04_pmc_summary.json profiles/r04_pmc_summary.json
python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err
python tools/debug/sha_shapes.py trained 2>&1 | grep -v amdgpu > gpurun_out/sha_last_trained.txt
cat $O
python -c "
import json; d=json.loads(open('gpurun_out/r04_bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'][:60], d['output_sha1'])"
