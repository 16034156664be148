#!/bin/bash
# SQ counters of the three hot kernels at the bench shape, for the current library and for a
# library built from the round-1 sources (folve_amd/libfolve_amd_r01.so, made by checking out the
# round-1 commit and `make -C folve_amd/csrc`).  Output: gpurun_out/sq_ba/{now,r01}.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/sq_ba
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_BRANCH")
for which in now r01; do
  if [ $which = r01 ]; then export FOLVE_AMD_LIB=$R/folve_amd/libfolve_amd_r01.so; else unset FOLVE_AMD_LIB; fi
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/$which/p$i -- python3 $R/tools/sq_bench.py > $OUT/$which.log$i 2>&1
  done
  python3 - <<PY > $OUT/$which.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/$which/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        key = None
        for k in ("mac_walk_kernel", "mac_slide_kernel", "forward_walker_kernel", "inverse_walker_kernel"):
            if k in n: key = k
        if not key: continue
        a = acc[key][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(k)
    for c, (s, n) in sorted(acc[k].items()):
        print("   %-28s %16.0f  (avg over %d dispatches)" % (c, s / n, n))
PY
  cat $OUT/$which.txt
done
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
