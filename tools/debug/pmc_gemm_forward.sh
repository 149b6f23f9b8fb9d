#!/bin/bash
# SQ counters of the linear-layer kernels INSIDE a 256 x 125 forward_offline (one counter group per rocprofv3 pass, kernel-trace only)
cd /tmp; export TMPDIR=/tmp
S=$GRAFT_REPO_ROOT/tools/debug/timeline.py
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  d=/tmp/pg_$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -o p -- python $S --child 256 125 > /dev/null 2>&1
  f=$(find $d -name "p_counter_collection.csv" | head -1)
  echo "=== $grp"
  python3 - "$f" <<'PY'
import csv, sys, collections
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("  (no data)", e); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
    if "gemm" not in k and "fused<256, 8, 512" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    print("  %-46s" % k, {c: "%.4g" % (v / cnt[(k, c)]) for c, v in acc[k].items()})
PY
done
