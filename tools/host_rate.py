"""PCIe-inclusive rate of the host-pointer batch path (dev aid; the number quoted in DESIGN.md §8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import folve_amd as fa
from folve_amd.capi import BatchPlan, FE_HOST_PTRS

S, T, P, size = 64, 32, 8192, 262144
eng = fa.Engine(0)
flt = fa.Filter(eng, 2, 2, size)
rng = np.random.default_rng(3)
for c in range(2):
    h = rng.standard_normal(size).astype(np.float32); h /= np.linalg.norm(h); flt.add(c, c, h)
flt.commit()
streams = [flt.open_stream(T) for _ in range(S)]
xs = [rng.uniform(-1, 1, (T * P, 2)).astype(np.float32) for _ in range(S)]
ys = [np.zeros_like(x) for x in xs]
plan = BatchPlan(streams, [x.ctypes.data for x in xs], [y.ctypes.data for y in ys], [T * P] * S, FE_HOST_PTRS)
for _ in range(2):
    plan.run()
t0 = time.perf_counter(); n = 5
for _ in range(n):
    plan.run()
dt = (time.perf_counter() - t0) / n
print("host-pointer batch: %.2f ms/step, %.1f Msamples/s (pageable host memory, H2D + K1-K3 + D2H)" % (dt * 1e3, S * T * P * 2 / dt / 1e6))

# the same with page-locked caller buffers
import torch
xp = [torch.from_numpy(x).pin_memory() for x in xs]
yp = [torch.zeros(T * P, 2).pin_memory() for _ in xs]
planp = BatchPlan(streams, [x.data_ptr() for x in xp], [y.data_ptr() for y in yp], [T * P] * S, FE_HOST_PTRS)
for _ in range(2):
    planp.run()
t0 = time.perf_counter()
for _ in range(n):
    planp.run()
dt = (time.perf_counter() - t0) / n
print("host-pointer batch: %.2f ms/step, %.1f Msamples/s (page-locked host memory)" % (dt * 1e3, S * T * P * 2 / dt / 1e6))

# one block per call through the exact SoundProcessor::Process path (host pointers, synchronous, peaks fetched)
st = flt.open_stream(1)
x = xs[0][:P]
for _ in range(40):
    st.process(x)
t0 = time.perf_counter(); n = 200
for _ in range(n):
    st.process(x)
dt = (time.perf_counter() - t0) / n
print("fe_stream_process, one stereo block per call: %.1f us/block = %.1f Mframes/s = %.0fx real time at 44.1 kHz" % (
    dt * 1e6, P / dt / 1e6, P / dt / 44100))
