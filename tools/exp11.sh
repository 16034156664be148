#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -X faulthandler -m pytest tests -q -m gpu --timeout=300 -p no:cacheprovider 2>&1 | tail -4
bash tools/profile.sh r02b > gpurun_out/profile_r02b.log 2>&1
tail -28 gpurun_out/profile_r02b.log
timeout 900 python bench.py > gpurun_out/bench_r02b.json 2> gpurun_out/bench_r02b.err
tail -c 2500 gpurun_out/bench_r02b.json; tail -3 gpurun_out/bench_r02b.err
