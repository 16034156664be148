#!/bin/bash
# Dev aid (round 5): what would three packed FMAs per two rows buy K2?  -DFOLVE_EXPERIMENT_CUT drops every other
# imaginary-part FMA of the walk (WRONG results, right instruction count) — base / cut / base on one box.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in base cut base; do
  echo "== $v"
  FOLVE_AMD_LIB=$PWD/folve_amd/variants/libfolve_amd_$v.so timeout 300 python tools/config_rates.py 256 2>&1 | grep -v "T= 32"
  FOLVE_AMD_LIB=$PWD/folve_amd/variants/libfolve_amd_$v.so timeout 300 python tools/matrix_rate.py 2>&1 | tail -2
done
