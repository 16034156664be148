"""Dev aid: four 8-channel streams x 256 blocks (K = 32), the shape DESIGN.md quotes for the channel-pair walkers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchlib.configs import measure_config
r = measure_config(T=256, tune=None, steps=100, check=False, S=4, C=8, size=262144)
print("4 x 8 ch x 256 blocks: %.3f ms/call, kernels %s, %.1f Msamples/s" % (r["ms_per_call"], {k: round(v, 3) for k, v in r["kernels_ms"].items()}, r["msamples_per_s"]))
