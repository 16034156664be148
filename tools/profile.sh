#!/bin/bash
# rocprofv3 evidence for bench.py: kernel-trace stats, then separate PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).  Run on the GPU box:
#   bash tools/profile.sh <tag> [bench args...]
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline $@"   # bench.py defaults: 20 warm-up + 500 timed steps
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/bench_trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py $ARGS > $OUT/bench_$c.log 2>&1
done
find $OUT -name "*.csv" | head -50
python3 $R/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep only small files for the merge back
find $OUT -name "*.csv" -size +3M -delete
find $OUT -name "*.db" -delete
