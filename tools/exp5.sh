#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -X faulthandler -m pytest tests/test_forms_gpu.py -q -m gpu --timeout=300 -p no:cacheprovider -k "mac_forms or bench" 2>&1 | tail -5
for tune in "mac_form=100" "mac_form=101" "mac_form=102" "mac_form=103"; do
  echo "=== QB_TUNE=$tune"
  QB_TUNE=$tune timeout 300 python tools/quick_bench.py 64 64 200 2>&1 | tail -4 | grep -E "S=|mac"
done
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sb -- python3 $R/tools/single_block.py 300 > $R/gpurun_out/sb.log 2>&1
tail -4 $R/gpurun_out/sb.log
find $R/gpurun_out/sb -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200
find $R/gpurun_out/sb -name "*.csv" -size +1M -delete; find $R/gpurun_out/sb -name "*.db" -delete
