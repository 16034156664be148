"""The drop-in call in isolation: fe_stream_process, one synchronous stereo block (K = 32), looped.
Prints wall time per call; under `rocprofv3 --kernel-trace --stats` the per-kernel share."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import folve_amd as fa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
size, C = 262144, 2
eng = fa.Engine(0)
if os.environ.get("QB_TUNE"):
    eng.set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in os.environ["QB_TUNE"].split(","))})
flt = fa.Filter(eng, C, C, size)
rng = np.random.default_rng(3)
for c in range(C):
    h = rng.standard_normal(size).astype(np.float32); h /= np.linalg.norm(h)
    flt.add(c, c, h)
flt.commit()
P = flt.block_size
L = fa.lib()
buf = ctypes.c_void_p()
assert L.fe_host_alloc(P * C * 4, ctypes.byref(buf)) == 0
st = flt.open_stream(1)
assert L.fe_stream_bind_host_buffer(st.h, buf, P * C * 4) == 0
arr = np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(ctypes.c_float)), shape=(P * C,))
arr[:] = rng.uniform(-1, 1, P * C).astype(np.float32)
for _ in range(40):
    L.fe_stream_process(st.h, buf, P, buf, None, None)
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(n):
        L.fe_stream_process(st.h, buf, P, buf, None, None)
    print("zero-copy: %.1f us per block" % ((time.perf_counter() - t0) / n * 1e6))
