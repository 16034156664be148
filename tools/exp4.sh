#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sq4
cd $R
timeout 900 python -X faulthandler -m pytest tests/test_forms_gpu.py tests/test_multi_gpu.py -q -m gpu --timeout=300 -p no:cacheprovider 2>&1 | tail -5
for tune in "mac_form=100" "mac_form=101" "mac_form=102" "mac_form=103" "mac_form=16"; do
  echo "=== QB_TUNE=$tune"
  QB_TUNE=$tune timeout 300 python tools/quick_bench.py 64 64 200 2>&1 | tail -4 | grep -E "S=|mac"
done
cd /tmp && export TMPDIR=/tmp
QB_TUNE=mac_form=101 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/sq4/p1 -- python3 $R/tools/quick_bench.py 64 64 20 > $R/gpurun_out/sq4/log1.txt 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$R/gpurun_out/sq4/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        key = None
        for k in ("mac_walk_kernel",):
            if k in n: key = k
        if not key: continue
        a = acc[key][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(k)
    for c, (s, n) in sorted(acc[k].items()):
        print("   %-28s %16.0f  (avg over %d dispatches)" % (c, s / n, n))
PY
find $R/gpurun_out/sq4 -name "*.csv" -size +1M -delete; find $R/gpurun_out/sq4 -name "*.db" -delete
