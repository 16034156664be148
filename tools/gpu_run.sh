#!/bin/bash
# Helper for gpurun calls: run pytest -m gpu with per-test timeouts, log to gpurun_out/.
mkdir -p gpurun_out
timeout ${1:-600} python -X faulthandler -m pytest tests -x -q -m gpu --timeout=120 -p no:cacheprovider "${@:2}" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
