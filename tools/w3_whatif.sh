#!/bin/bash
# Dev aid (round 5): where the three-FMA walk's time goes — the product kernels, the walk without its stores, without
# loads and stores (tools/build_variant.sh w3nostore -DFOLVE_W3_NOSTORE, w3nomem -DFOLVE_W3_NOLOAD -DFOLVE_W3_NOSTORE).
cd "$(dirname "$0")/.."
for v in w3base w3nostore w3nomem w3base; do
  echo "== $v"
  FOLVE_AMD_LIB=$PWD/folve_amd/variants/libfolve_amd_$v.so timeout 300 python tools/config_rates.py 256 walk_fma=3 2>&1 | grep -v "T= 32\|amdgpu.ids"
  FOLVE_AMD_LIB=$PWD/folve_amd/variants/libfolve_amd_$v.so timeout 300 python tools/matrix_rate.py walk_fma=3 2>&1 | tail -1
done
