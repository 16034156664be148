"""Dev aid: the engine's automatic form choices over batch shapes — streams x blocks per call at 262 144 taps, stereo —
rate, per-kernel dispatch times and the kernels chosen: a rate that falls when the batch grows is a cliff in a threshold.
usage: python tools/sweep_shapes.py [taps] [channels]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchlib.configs import measure_config

size = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
C = int(sys.argv[2]) if len(sys.argv) > 2 else 2
print("# taps %d, %d channels; Gsamples/s (ms per call) [K1 K2 K3 us] kernels" % (size, C))
for T in (8, 16, 64, 256):
    for S in (1, 2, 4, 8, 16, 32, 64, 128):
        if S * T * C * 8192 * 8 * 3 > 24e9:
            continue
        try:
            r = measure_config(S=S, C=C, size=size, T=T, steps=40, warmup=5, check=False)
        except Exception as e:  # noqa: BLE001
            print("S=%3d T=%3d: %r" % (S, T, e)); continue
        k = r["kernels_ms"]; n = r["kernels_launched"]
        print("S=%3d T=%3d: %7.1f (%8.4f ms) [%6.1f %6.1f %6.1f]  %s | %s | %s" % (
            S, T, r["msamples_per_s"] / 1e3, r["ms_per_call"], k["forward"] * 1e3, k["mac"] * 1e3, k["inverse"] * 1e3,
            n["forward"].split("_kernel")[0] + n["forward"].split("_kernel")[1], n["mac"].replace("_kernel", ""), n["inverse"].split("_kernel")[0] + n["inverse"].split("_kernel")[1]), flush=True)
