"""Dev aid: a lone stereo stream's 256-block calls at filter lengths that walk every rung of the one-lane window ladder
(9 / 13 / 17 / 21 / 26 / 29 / 33 rows) — per-kernel times; run under FOLVE_AMD_LIB=<variant> to compare builds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchlib.configs import measure_config
for size in (65536, 98304, 131072, 163840, 204800, 229376, 262144):
    r = measure_config(S=1, C=2, size=size, T=256, tune=None, steps=300, check=False)
    print("size %7d  %.4f ms/call  K2 %s  %s" % (size, r["ms_per_call"], r["kernels_launched"]["mac"], {k: round(v * 1e3, 1) for k, v in r["kernels_ms"].items()}), flush=True)
