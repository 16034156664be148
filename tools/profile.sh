#!/bin/bash
# rocprofv3 evidence for bench.py: kernel-trace stats, then separate PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).  Run on the GPU box:
#   bash tools/profile.sh <tag> [bench args...]
# Leaves gpurun_out/prof_<tag>/{summary.json,kernel_stats.csv}; copy them to profiles/<tag>_*.
TAG=${1:-r02}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --skip drop_in,configs,end_to_end,single_block,mixed $@"   # bench.py defaults: 20 warm-up + 500 timed steps; the legs left out run child processes or other shapes
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/bench_trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py $ARGS > $OUT/bench_$c.log 2>&1
done
python3 $R/tools/summarize_profile.py $OUT > $OUT/summary.json 2> $OUT/summary.err
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
tail -3 $OUT/bench_trace.log | cut -c1-600
python3 - <<PY
import json
s = json.load(open("$OUT/summary.json"))
for k, v in sorted(s["kernel_trace"].items()): print("%-60s n=%5d avg %9.1f us" % (k[:60], v["dispatches"], v["avg_ns"] / 1e3))
for k, v in sorted(s["hbm_per_dispatch"].items()): print("%-60s R %8.1f MB  W %8.1f MB" % (k[:60], v["hbm_read_bytes_corrected"] / 1e6, v["hbm_write_bytes"] / 1e6))
PY
# keep only small files for the merge back
find $OUT -name "*.csv" -size +3M -delete
find $OUT -name "*.db" -delete
