"""Dev aid (GPU): where the stereo block walkers overtake the per-channel general kernels of K1 / K3 — streams x blocks per
call against ms per call and per-kernel ms, automatic choice vs fft_form = 1 (general kernels only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for S, T in ((1, 128), (1, 256), (1, 384), (1, 512), (2, 256), (3, 256), (4, 256), (1, 1024), (8, 64), (8, 128)):
    row = []
    for tune in (None, {"fft_form": 1}):
        r = bench.measure_config(S=S, C=2, size=204800, T=T, populated=178193, steps=60, warmup=8, tune=tune, check=False)
        row.append((round(r["ms_per_call"] * 1e3, 1), {k: round(v * 1e3, 1) for k, v in r["kernels_ms"].items()}))
    print("S=%d T=%4d  walkers %s   general %s" % (S, T, row[0], row[1]), flush=True)
