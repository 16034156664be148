#!/bin/bash
# Dev aid: A/B/A/B of two libraries of folve_amd/variants (tools/build_variant.sh) over cfg2 / cfg4 / MAXSIZE / cfg3 and the
# 2 x 2 matrix.  usage: tools/ab_variants.sh nameA nameB
cd "$(dirname "$0")/.."
for v in $1 $2 $1 $2; do
  echo "== $v"
  FOLVE_AMD_LIB=$PWD/folve_amd/variants/libfolve_amd_$v.so timeout 300 python tools/config_rates.py 256 2>&1 | grep -v "T= 32\|amdgpu.ids"
  FOLVE_AMD_LIB=$PWD/folve_amd/variants/libfolve_amd_$v.so timeout 300 python tools/matrix_rate.py 2>&1 | tail -1
done
