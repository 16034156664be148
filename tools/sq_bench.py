"""The bench shape (64 streams x stereo x 64 blocks, 262 144 taps), 20 steps, through the plain C ABI
only — so that it also runs against a library built from older sources (FOLVE_AMD_LIB)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

lib = C.CDLL(os.environ.get("FOLVE_AMD_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "folve_amd", "libfolve_amd.so"))
vp = C.c_void_p
S, T, P, size = 64, 64, 8192, 262144
eng = vp(); assert lib.fe_engine_create(0, None, C.byref(eng)) == 0
flt = vp(); assert lib.fe_filter_create(eng, 2, 2, size, C.c_float(0), C.byref(flt)) == 0
rng = np.random.default_rng(3)
for c in range(2):
    h = rng.standard_normal(size).astype(np.float32); h /= np.linalg.norm(h)
    assert lib.fe_filter_add(flt, c, c, 1, h.ctypes.data_as(vp), 0, size) == 0
assert lib.fe_filter_commit(flt) == 0
streams = []
for _ in range(S):
    s = vp(); assert lib.fe_stream_open(flt, T, C.byref(s)) == 0
    streams.append(s)
xs = [torch.rand(T * P, 2, device="cuda") * 2 - 1 for _ in range(S)]
ys = [torch.empty_like(x) for x in xs]
torch.cuda.synchronize()
sa = (vp * S)(*[s.value for s in streams])
ia = (vp * S)(*[x.data_ptr() for x in xs])
oa = (vp * S)(*[y.data_ptr() for y in ys])
na = (C.c_longlong * S)(*([T * P] * S))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    assert lib.fe_batch_process(sa, S, ia, na, oa, 3) == 0     # FE_DEVICE_PTRS | FE_ASYNC
assert lib.fe_engine_synchronize(eng) == 0
print("ok")
