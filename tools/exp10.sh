#!/bin/bash
cd $GRAFT_REPO_ROOT
for tune in "mac_form=100" "mac_form=102" "mac_form=103"; do
  echo "=== QB_TUNE=$tune"
  QB_TUNE=$tune timeout 300 python tools/quick_bench.py 64 64 300 2>&1 | tail -4 | grep -E "S=|mac"
done
