#!/bin/bash
# What the driver runs at round end, in one gpurun call: pytest -m gpu, smoke, the bench line.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -X faulthandler -m pytest tests -q -m gpu --timeout=300 -p no:cacheprovider 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()"
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_like.json 2> gpurun_out/bench_driver_like.err
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_driver_like.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "steady", d["steady_state"], "parity", d["parity_rms"])
print("roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic", "traffic_source", "traffic_note", "kernel_ms", "kernels_ms")})
print("single", d["single_block"]); print("e2e", d["end_to_end"]); print("cpu", d["cpu_baseline"])
PY
tail -3 gpurun_out/bench_driver_like.err
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_driver_like.json").read().strip().splitlines()[-1])
print("drop_in", json.dumps(d.get("drop_in_threads"))[:900])
PY
