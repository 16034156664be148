#!/bin/bash
# All the rocprofv3 evidence bench.py's roofline blocks rest on, in one GPU call: cfg3 (the default bench command), cfg1,
# cfg2, cfg4, the 2 x 2 matrix (`--only-config`) — kernel trace + the two PMC passes each (tools/profile.sh).  usage: tools/profile_all.sh r05
TAG=${1:-r05}
cd "$(dirname "$0")/.."
bash tools/profile.sh $TAG
for c in cfg1 cfg2 cfg4 matrix; do bash tools/profile.sh ${TAG}_$c --only-config $c --steps 300 --skip longer; done
