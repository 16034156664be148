#!/bin/bash
# Dev aid: cfg4 (one 8-channel stream, K = 64, 256-block calls) over the K1 / K3 walker run lengths.
cd "$(dirname "$0")/.."
for r in 0 1 2 4 8; do
  echo "== fwd_run=$r inv_run=$r"
  timeout 300 python tools/config_rates.py 256 fwd_run=$r,inv_run=$r 2>&1 | grep "cfg4 .*T=256"
done
for f in 1 3; do
  echo "== fft_form=$f"
  timeout 300 python tools/config_rates.py 256 fft_form=$f 2>&1 | grep "cfg4 .*T=256"
done
