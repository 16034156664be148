#!/bin/bash
# Dev aid: socket power and shader clock (amdgpu hwmon) while tools/quick_bench.py loops.
# usage: clock_watch.sh T steps  (FOLVE_AMD_LIB selects a variant library)
cd "$(dirname "$0")/.."
T=${1:-256}; STEPS=${2:-5000}
python tools/quick_bench.py 64 $T $STEPS > gpurun_out/clock_watch_bench.log 2>&1 &
pid=$!
sleep 8
for i in $(seq 1 10); do
  best=0; line=""
  for h in /sys/class/drm/card*/device/hwmon/hwmon*; do
    [ -r $h/power1_input ] || continue
    p=$(cat $h/power1_input); f=$(cat $h/freq1_input)
    if [ "$p" -gt "$best" ]; then best=$p; line="power $((p/1000000)) W sclk $((f/1000000)) MHz"; fi
  done
  echo "$line"
  sleep 0.25
done
wait $pid
tail -4 gpurun_out/clock_watch_bench.log
