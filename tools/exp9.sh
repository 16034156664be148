#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -X faulthandler -m pytest tests/test_forms_gpu.py -q -m gpu --timeout=300 -p no:cacheprovider -k "mac_forms or bench" 2>&1 | tail -3
for tune in "mac_form=100" "mac_form=101"; do
  echo "=== QB_TUNE=$tune"
  QB_TUNE=$tune timeout 300 python tools/quick_bench.py 64 64 300 2>&1 | tail -4 | grep -E "S=|mac"
done
