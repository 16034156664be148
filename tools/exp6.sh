#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -X faulthandler -m pytest tests -q -m gpu --timeout=300 -p no:cacheprovider -x 2>&1 | tail -5
python tools/single_block.py 300 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sb2 -- python3 $R/tools/single_block.py 300 > $R/gpurun_out/sb2.log 2>&1
find $R/gpurun_out/sb2 -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-150 | head -5
find $R/gpurun_out/sb2 -name "*.csv" -size +1M -delete; find $R/gpurun_out/sb2 -name "*.db" -delete
