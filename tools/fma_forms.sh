#!/bin/bash
# Dev aid (round 5): K2's whole-call walk with four against three FMAs per complex multiply-add, A/B/A on one box:
# cfg2 / cfg4 / MAXSIZE / cfg3 (tools/config_rates.py) and cfg3's batch with a 2 x 2 matrix (tools/matrix_rate.py).
cd "$(dirname "$0")/.."
for v in 4 3 4 3; do
  echo "== walk_fma=$v"
  timeout 300 python tools/config_rates.py 256 walk_fma=$v 2>&1 | grep -v "T= 32\|amdgpu.ids"
  timeout 300 python tools/matrix_rate.py walk_fma=$v 2>&1 | tail -2
done
