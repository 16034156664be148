#!/bin/bash
# A short sweep of the harness (many-thread cases).   tools/dropin/sweep_short.sh [out file]
out=${1:-gpurun_out/dropin_sweep_short.txt}
conf=$(python3 tools/dropin/make_conf.py /tmp/folve_dropin_conf)
exe=tools/dropin/dropin_threads
{
  for spec in "1 20000 1 32" "4 8000 1 32" "16 4096 1 32" "16 4096 1 128" "64 300 1 1" "64 2048 1 8" "64 2048 1 32" "64 2048 1 64" "64 2048 1 128" "128 1024 1 32"; do
    set -- $spec
    timeout 300 $exe "$conf" $1 $2 $3 run_ahead=$4 || echo "FAILED: $spec"
  done
} 2>&1 | tee "$out"
