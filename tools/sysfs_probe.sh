#!/bin/bash
cd "$(dirname "$0")/.."
python tools/quick_bench.py 64 256 3000 > /dev/null 2>&1 &
pid=$!
sleep 9
for d in /sys/class/drm/card*/device; do
  echo "== $d"; cat $d/vendor 2>/dev/null
  for h in $d/hwmon/hwmon*; do
    for f in $h/power1_average $h/power1_input $h/power1_cap $h/freq1_input $h/freq2_input $h/temp1_input; do [ -r $f ] && echo "$f: $(cat $f)"; done
  done
  [ -r $d/pp_dpm_sclk ] && { echo pp_dpm_sclk; cat $d/pp_dpm_sclk; }
  [ -r $d/gpu_busy_percent ] && echo "busy $(cat $d/gpu_busy_percent)"
done 2>&1 | head -60
wait $pid
