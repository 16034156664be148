#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -X faulthandler -m pytest tests -q -m gpu --timeout=300 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
for tune in "mac_form=100" "mac_form=101" "mac_form=102" "mac_form=103" "mac_form=16"; do
  echo "=== QB_TUNE=$tune"
  QB_TUNE=$tune timeout 300 python tools/quick_bench.py 64 64 200 2>&1 | tail -4
done
echo "=== bench default"; timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernels_ms'])"
echo "=== bench mac16"; timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --tune mac_form=16 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernels_ms'])"
# SQ counters on the three hot kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/sq/p$i -- python3 $R/tools/quick_bench.py 64 64 20 > $R/gpurun_out/sq/log$i.txt 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$R/gpurun_out/sq/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        key = None
        for k in ("mac_walk_kernel", "forward_walker_kernel", "inverse_walker_kernel"):
            if k in n: key = k
        if not key: continue
        a = acc[key][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(k)
    for c, (s, n) in sorted(acc[k].items()):
        print("   %-28s %16.0f  (avg over %d dispatches)" % (c, s / n, n))
PY
find $R/gpurun_out/sq -name "*.csv" -size +1M -delete; find $R/gpurun_out/sq -name "*.db" -delete
