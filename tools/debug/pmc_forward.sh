#!/bin/bash
# HBM bytes per launch of every kernel of ONE 256 x 125 forward_offline (exact-fp32 mode), from separate rocprofv3 --pmc passes
# (FETCH_SIZE, WRITE_SIZE; bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 as MI355X_MICROARCH.md prescribes) plus the kernel durations
# of a --kernel-trace --stats pass -> GB/s per kernel.  Unlike tools/profile.py (which profiles bench.py with its 1024-sequence leg)
# every launch here has the BASELINE shape, so bytes per launch compare directly with the algorithmic figures in DESIGN.md.
cd /tmp; export TMPDIR=/tmp
S=$GRAFT_REPO_ROOT/tools/debug/timeline.py
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pf_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pf_$c -o p -- python $S --child 256 125 > /dev/null 2>&1
done
rm -rf /tmp/pf_t; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_t -o p -- python $S --child 256 125 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
def load(c):
    f = glob.glob("/tmp/pf_%s/**/p_counter_collection.csv" % c, recursive=True)[0]
    acc, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        acc[k] += float(r["Counter_Value"]); n[k] += 1
    return {k: acc[k] / n[k] for k in acc}
fe, wr = load("FETCH_SIZE"), load("WRITE_SIZE")
st = {}
for r in csv.DictReader(open(glob.glob("/tmp/pf_t/**/p_kernel_stats.csv", recursive=True)[0])):
    st[r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]] = float(r["AverageNs"]) / 1e3
print("%-44s %9s %9s %9s %8s %8s" % ("kernel (256 x 125 forward_offline)", "fetch MB", "write MB", "HBM MB", "avg us", "TB/s"))
for k in sorted(st, key=lambda k: -st[k]):
    if not k.startswith("mp_") or k.startswith("mp_pack") or k not in fe: continue
    f, w = fe[k] * 1024 / 1e6, wr.get(k, 0) * 1024 / 1e6
    tot = 2 * f + w
    print("%-44s %9.1f %9.1f %9.1f %8.1f %8.2f" % (k[:44], f, w, tot, st[k], tot / st[k]))
PY
