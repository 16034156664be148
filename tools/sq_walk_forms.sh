#!/bin/bash
# SQ counters of K2's whole-call walk with four and with three FMAs per complex multiply-add, at cfg4's shape (one
# 8-channel stream, K = 64, 256-block calls: two lanes per bin, instruction-issue bound) and at cfg3's (64 stereo streams:
# memory bound).  Separate --pmc passes, no tracing beside them.  Output: gpurun_out/sq_walk/{cfg4,cfg3}_fma{4,3}.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/sq_walk
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU")
for shape in cfg4 cfg3; do
for fma in 4 3; do
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/raw_${shape}_$fma/p$i -- python3 $R/tools/sq_walk_bench.py $shape $fma > $OUT/${shape}_$fma.log$i 2>&1
  done
  python3 - <<PY > $OUT/${shape}_fma$fma.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/raw_${shape}_$fma/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        key = None
        for k in ("mac_walk3_nt_kernel", "mac_walk3_kernel", "mac_walk_kernel"):
            if k in n and key is None: key = k
        if not key: continue
        a = acc[key][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
print("$shape, walk_fma=$fma  (rocprofv3 --pmc, averages per dispatch)")
for k in sorted(acc):
    print(k)
    d = {c: s / n for c, (s, n) in acc[k].items()}
    for c in sorted(d):
        print("   %-28s %16.0f" % (c, d[c]))
    if d.get("SQ_WAVES") and d.get("SQ_WAVE_CYCLES"):
        w = d["SQ_WAVES"]
        print("   per wavefront: %.0f vector + %.0f scalar + %.0f memory instructions; of its cycles %.0f %% issuing, %.0f %% waiting to issue, %.0f %% parked at s_waitcnt"
              % (d.get("SQ_INSTS_VALU", 0) / w, d.get("SQ_INSTS_SALU", 0) / w, d.get("SQ_INSTS_VMEM", 0) / w,
                 100 * d.get("SQ_ACTIVE_INST_ANY", 0) / d["SQ_WAVE_CYCLES"], 100 * d.get("SQ_WAIT_INST_ANY", 0) / d["SQ_WAVE_CYCLES"],
                 100 * d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"]))
PY
  cat $OUT/${shape}_fma$fma.txt
done
done
rm -rf $OUT/raw_*
