#!/bin/bash
# round-2 first GPU pass: new tests, then kernel-form experiments at the bench shape
mkdir -p gpurun_out
timeout 1500 python -X faulthandler -m pytest tests -x -q -m gpu --timeout=300 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
for tune in "" "mac_form=16" "fft_form=1" "fwd_run=4,inv_run=4" "fwd_run=16,inv_run=16" "fwd_run=32,inv_run=32"; do
  echo "=== QB_TUNE=$tune"
  QB_TUNE=$tune timeout 300 python tools/quick_bench.py 64 64 200 2>&1 | tail -5
done
timeout 600 python bench.py --steps 200 --warmup 20 > gpurun_out/bench1.json 2> gpurun_out/bench1.err
tail -c 3000 gpurun_out/bench1.json; tail -5 gpurun_out/bench1.err
