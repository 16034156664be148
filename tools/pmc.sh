#!/bin/bash
# SQ counter passes over tools/quick_bench.py (dev aid).  usage: bash tools/pmc.sh <tag> "<counters pass 1>" "<counters pass 2>" ...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/quick_bench.py 64 32 3 > $OUT/log$i.txt 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        key = None
        for k in ("forward_kernel", "mac_slide_kernel", "mac_kernel", "inverse_kernel", "forward_walker_kernel", "inverse_walker_kernel", "forward_dual_kernel"):
            if k in n: key = k
        if not key: continue
        key += " grid=" + row["Grid_Size"]
        a = acc[key][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(k)
    for c, (s, n) in sorted(acc[k].items()):
        print("   %-28s %16.0f  (avg over %d dispatches)" % (c, s / n, n))
PY
find $OUT -name "*.csv" -size +2M -delete
