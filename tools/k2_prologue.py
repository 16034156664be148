"""Dev aid: K2's time against the blocks per call of a lone stream — the intercept is the walk's prologue (G and the history:
33 + 33 rows per lane, a burst of HBM reads in front of the first step, nothing to overlap it with), the slope one step.
usage: python tools/k2_prologue.py [cfg4|cfg2|cfg1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from benchlib.configs import OTHER_CONFIGS, measure_config

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
cfg = dict(OTHER_CONFIGS[name]); cfg.pop("frames", None)
ts, k1, k2, k3, wall = [], [], [], [], []
for T in (32, 64, 128, 256, 512, 1024):
    r = measure_config(T=T, steps=60, check=False, **cfg)
    ts.append(T); k1.append(r["kernels_ms"]["forward"] * 1e3); k2.append(r["kernels_ms"]["mac"] * 1e3); k3.append(r["kernels_ms"]["inverse"] * 1e3)
    wall.append(r["ms_per_call"] * 1e3)
    print("%s T=%4d: call %7.1f us  K1 %6.1f  K2 %6.1f  K3 %6.1f  (%s)" % (name, T, wall[-1], k1[-1], k2[-1], k3[-1], r["kernels_launched"]["mac"]))
for lbl, v in (("K1", k1), ("K2", k2), ("K3", k3), ("call", wall)):
    a, b = np.polyfit(ts[2:], v[2:], 1)
    print("%-4s = %6.2f us + %.4f us per block (fit over T >= 128)" % (lbl, b, a))
