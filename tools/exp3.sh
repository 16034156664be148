#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sq
cd $R
timeout 600 python -X faulthandler -m pytest tests/test_multi_gpu.py tests/test_host_gpu.py -q -m gpu --timeout=300 -p no:cacheprovider 2>&1 | tail -5
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['single_block'], d['roofline_streaming']['kernels_ms'], d['end_to_end'])"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  QB_TUNE=mac_form=101 timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/sq/p$i -- python3 $R/tools/quick_bench.py 64 64 20 > $R/gpurun_out/sq/log$i.txt 2>&1
  tail -2 $R/gpurun_out/sq/log$i.txt
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
