#!/bin/bash
# The drop-in harness over thread counts and run-ahead depths (run on the GPU box from the repo root).
#   tools/dropin/sweep.sh [out file]
out=${1:-gpurun_out/dropin_sweep.txt}
mkdir -p "$(dirname "$out")"
conf=$(python3 tools/dropin/make_conf.py /tmp/folve_dropin_conf)
exe=tools/dropin/dropin_threads
{
  nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
  for spec in "1 300 1 1" "1 20000 1 8" "1 20000 1 32" "1 20000 1 128" "4 8000 1 32" "16 300 1 1" "16 4096 1 32" "16 4096 1 128" \
              "64 300 1 1" "64 2048 1 8" "64 2048 1 32" "64 2048 1 128" "64 300 0 1" "64 2048 0 32" "128 1024 1 32"; do
    set -- $spec
    timeout 300 $exe "$conf" $1 $2 $3 run_ahead=$4 || echo "FAILED: $spec"
  done
} 2>&1 | tee "$out"
