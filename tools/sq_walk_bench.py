"""One shape's run-ahead calls with the walk pinned to four or three FMAs per complex MAC (tools/sq_walk_forms.sh profiles
this).  usage: sq_walk_bench.py cfg4|cfg3 4|3"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import folve_amd as fa
from folve_amd.capi import BatchPlan, FE_ASYNC, FE_DEVICE_PTRS

shape, fma = sys.argv[1], int(sys.argv[2])
S, C, size, T = (1, 8, 524288, 256) if shape == "cfg4" else (64, 2, 262144, 256)
ts = torch.cuda.Stream()
eng = fa.Engine(0, ts.cuda_stream)
eng.set_tuning(walk_fma=fma)
flt = fa.Filter(eng, C, C, size)
rng = np.random.default_rng(3)
for c in range(C):
    h = rng.standard_normal(size).astype(np.float32)
    flt.add(c, c, h / np.linalg.norm(h))
flt.commit()
P = flt.block_size
streams = [flt.open_stream(T) for _ in range(S)]
with torch.cuda.stream(ts):
    xs = [torch.rand(T * P, C, device="cuda") * 2 - 1 for _ in range(S)]
    ys = [torch.empty_like(x) for x in xs]
plan = BatchPlan(streams, [x.data_ptr() for x in xs], [y.data_ptr() for y in ys], [T * P] * S, FE_DEVICE_PTRS | FE_ASYNC)
for _ in range(12):
    plan.run()
eng.synchronize()
print("ok", eng.last_kernels()["mac"])
