#!/bin/bash
# What the driver runs at round end, in one gpurun call: pytest -m gpu, smoke, the bench line.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 2400 python -X faulthandler -m pytest tests -q -m gpu --timeout=600 -p no:cacheprovider 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke()"
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --details gpurun_out/bench_driver_like_details.json > gpurun_out/bench_driver_like.json 2> gpurun_out/bench_driver_like.err
python - <<PY
import json
text = open("gpurun_out/bench_driver_like.json").read().strip().splitlines()
d = json.loads(text[-1])
print("stdout lines", len(text), "line bytes", len(text[-1]))
print(text[-1])
PY
tail -c 600 gpurun_out/bench_driver_like.err | head -c 300
