"""Quick per-kernel timing at the cfg3 shape (64 streams x stereo, 256k taps). Dev aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import folve_amd as fa

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
size = 262144
P = 8192
ts = torch.cuda.Stream()
eng = fa.Engine(0, ts.cuda_stream)
if os.environ.get("QB_TUNE"):                    # e.g. QB_TUNE=mac_form=16,fwd_run=8
    eng.set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in os.environ["QB_TUNE"].split(","))})
rng = np.random.default_rng(3)
flt = fa.Filter(eng, 2, 2, size)
for c in range(2):
    h = rng.standard_normal(size).astype(np.float32); h /= np.linalg.norm(h)
    flt.add(c, c, h)
flt.commit()
streams = [flt.open_stream(T) for _ in range(S)]
with torch.cuda.stream(ts):
    xs = [torch.rand(T * P, 2, device="cuda") * 2 - 1 for _ in range(S)]
    ys = [torch.empty_like(x) for x in xs]
    if os.environ.get("QB_SAME_INPUT"):          # probe: every stream reads the same (cache-resident) PCM
        xs = [xs[0]] * S
from folve_amd.capi import BatchPlan, FE_DEVICE_PTRS, FE_ASYNC
plan = BatchPlan(streams, [x.data_ptr() for x in xs], [y.data_ptr() for y in ys], [T * P] * S, FE_DEVICE_PTRS | FE_ASYNC)
for _ in range(3):
    plan.run()
eng.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    plan.run()
eng.synchronize()
dt = (time.perf_counter() - t0) / steps
units = S * 2 * T
print("S=%d T=%d: %.3f ms/step, %.1f Msamples/s, %.2f M block-ch/s, alg %.2f TB/s" % (
    S, T, dt * 1e3, units * P / dt / 1e6, units / dt / 1e6, units * 2294028 / dt / 1e12))
eng.set_profiling(True); eng.reset_profile()
for _ in range(steps):
    plan.run()
eng.synchronize()
pr = eng.get_profile()
for k, v in pr.items():
    print("  %-8s %.3f ms/launch" % (k, v["ms"] / max(1, v["launches"])))
