"""Where the FFT kernels spend their cycles (dev aid; default: the cfg3 shape).
    python tools/phase_trace.py [streams] [blocks per call] [steps] [channels] [taps]

Needs the TRACE build:  make -C folve_amd/csrc TRACE=1   (libfolve_amd_trace.so)
The instrumented kernels add up, over workgroups, the shader-clock cycles wave 0 spends in
each phase; this prints the shares.  The product library carries none of this.
"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("FOLVE_AMD_LIB", os.path.join(ROOT, "folve_amd", "libfolve_amd_trace.so"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import folve_amd as fa
from folve_amd import capi
from folve_amd.capi import BatchPlan, FE_DEVICE_PTRS, FE_ASYNC

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
C = int(sys.argv[4]) if len(sys.argv) > 4 else 2
size, P = (int(sys.argv[5]) if len(sys.argv) > 5 else 262144), 8192
ts = torch.cuda.Stream()
eng = fa.Engine(0, ts.cuda_stream)
rng = np.random.default_rng(3)
flt = fa.Filter(eng, C, C, size)
for c in range(C):
    h = rng.standard_normal(size).astype(np.float32); h /= np.linalg.norm(h)
    flt.add(c, c, h)
flt.commit()
streams = [flt.open_stream(T) for _ in range(S)]
with torch.cuda.stream(ts):
    xs = [torch.rand(T * P, C, device="cuda") * 2 - 1 for _ in range(S)]
    ys = [torch.empty_like(x) for x in xs]
plan = BatchPlan(streams, [x.data_ptr() for x in xs], [y.data_ptr() for y in ys], [T * P] * S, FE_DEVICE_PTRS | FE_ASYNC)
L = capi.lib()
L.fe_debug_phases.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
for _ in range(3):
    plan.run()
eng.synchronize()
assert L.fe_debug_phases(buf, 1) == 0
for _ in range(steps):
    plan.run()
eng.synchronize()
assert L.fe_debug_phases(buf, 1) == 0
names = [
    ("forward_walker", ["PCM wait + stage A", "barrier", "stage B", "barrier", "split + stores", "barrier"]),
    ("inverse_walker", ["Y wait + fold", "prefetch + stage A", "barrier", "stage B", "barrier", "read + stores", "barrier"]),
]
for k, (kn, ph) in enumerate(names):
    v = [buf[k * 8 + i] for i in range(len(ph))]
    tot = float(sum(v)) or 1.0
    print("%s: %.0f cycles per workgroup-launch (sum over phases / launches)" % (kn, tot / steps))
    for n, c in zip(ph, v):
        print("   %-20s %5.1f %%" % (n, 100.0 * c / tot))
