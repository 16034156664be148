#!/bin/bash
# Dev aid: A/B/.. of libraries of folve_amd/variants (tools/build_variant.sh) on cfg3's 64 x 2 x 256 batch only: per-kernel
# dispatch times and the call, two rounds.  usage: tools/ab_cfg3.sh nameA nameB [nameC ..]
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for v in "$@"; do
    echo "== $v"
    FOLVE_AMD_LIB=$PWD/folve_amd/variants/libfolve_amd_$v.so timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0, ".")
from benchlib.configs import measure_config
r = measure_config(S=64, C=2, size=262144, T=256, steps=150, warmup=20, check=False)
k = r["kernels_ms"]
print("cfg3: %.4f ms/call  K1 %.4f K2 %.4f K3 %.4f  sum %.4f  %.1f Gsamples/s  sclk %s W %s" % (r["ms_per_call"], k["forward"], k["mac"], k["inverse"], sum(k.values()), r["msamples_per_s"] / 1e3, (r["power"] or {}).get("sclk_mhz"), (r["power"] or {}).get("socket_w")))
PY
  done
done
