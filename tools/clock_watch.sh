#!/bin/bash
# Dev aid: sample the shader clock and socket power while tools/quick_bench.py runs (is the step power-limited?).
cd "$(dirname "$0")/.."
T=${1:-256}
python tools/quick_bench.py 64 $T 5000 > gpurun_out/clock_watch_bench.log 2>&1 &
pid=$!
sleep 9
for i in $(seq 1 30); do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' '; echo
  sleep 0.4
done
wait $pid
tail -4 gpurun_out/clock_watch_bench.log
echo "== idle"
sleep 2
/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' '; echo
