#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -X faulthandler -m pytest tests -q -m gpu --timeout=300 -p no:cacheprovider -x 2>&1 | tail -5
python tools/single_block.py 300 | tail -2
QB_TUNE= timeout 300 python tools/quick_bench.py 64 64 200 2>&1 | tail -4
timeout 300 python tools/quick_bench.py 64 1 200 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sb3 -- python3 $R/tools/single_block.py 300 > $R/gpurun_out/sb3.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob('$R/gpurun_out/sb3/runc/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:70].replace('void fk::(anonymous namespace)::',''), r['Calls'], r['AverageNs'], r['MinNs'])
PY
find $R/gpurun_out/sb3 -name "*.csv" -size +1M -delete; find $R/gpurun_out/sb3 -name "*.db" -delete
