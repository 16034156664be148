#!/bin/bash
# Dev aid: tools/quick_bench.py over every library in folve_amd/variants (built by tools/build_variant.sh).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=${1:-256}; STEPS=${2:-60}
for rep in 1 2; do
for so in folve_amd/variants/libfolve_amd_*.so; do
  echo "== $(basename $so) rep $rep"
  FOLVE_AMD_LIB=$PWD/$so timeout 300 python tools/quick_bench.py 64 $T $STEPS 2>&1 | tail -5
done
done
