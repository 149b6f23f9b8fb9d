#!/bin/bash
# PMC passes over tools/micro/gemm_bench (GPU box): one counter group per pass, kernel-trace only.
cd /tmp; export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/tools/micro/gemm_bench
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA"; do
  d=/tmp/pmc_$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -o p -- $B > /dev/null 2>&1
  f=$(find $d -name "p_counter_collection.csv" | head -1)
  echo "=== $grp"
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:40]
    k += " g%s" % r.get("Grid_Size", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    print("  %-60s" % k, {c: "%.3g" % (v / cnt[(k, c)]) for c, v in acc[k].items()})
PY
done
