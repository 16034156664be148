#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
bash tools/profile.sh r02a > gpurun_out/profile_r02a.log 2>&1
tail -25 gpurun_out/profile_r02a.log
bash tools/sq_before_after.sh > gpurun_out/sq_ba.log 2>&1
tail -5 gpurun_out/sq_ba.log
