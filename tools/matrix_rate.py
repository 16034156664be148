"""Dev aid: cfg3-like batch with a FULL 2 x 2 filter matrix (four paths: true-stereo reverbs) against the diagonal one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import folve_amd as fa
from folve_amd.capi import BatchPlan, FE_DEVICE_PTRS, FE_ASYNC

def run(full, S=64, T=256, size=262144, tune=None):
    ts = torch.cuda.Stream(); eng = fa.Engine(0, ts.cuda_stream)
    if tune: eng.set_tuning(**tune)
    flt = fa.Filter(eng, 2, 2, size); rng = np.random.default_rng(3)
    for i in range(2):
        for o in range(2):
            if full or i == o:
                h = rng.standard_normal(size).astype(np.float32); flt.add(i, o, h / np.linalg.norm(h) * 0.5)
    flt.commit(); P = flt.block_size
    st = [flt.open_stream(T) for _ in range(S)]
    with torch.cuda.stream(ts):
        xs = [torch.rand(T * P, 2, device="cuda") * 2 - 1 for _ in range(S)]; ys = [torch.empty_like(x) for x in xs]
    plan = BatchPlan(st, [x.data_ptr() for x in xs], [y.data_ptr() for y in ys], [T * P] * S, FE_DEVICE_PTRS | FE_ASYNC)
    for _ in range(5): plan.run()
    eng.synchronize(); eng.reset_profile(); eng.set_profiling(2)
    for _ in range(30): plan.run()
    eng.synchronize(); p = eng.get_kernel_profile(); eng.set_profiling(0)
    k = {n: v["ms"] / v["launches"] for n, v in p.items()}
    print("full matrix" if full else "diagonal   ", {n: round(v, 3) for n, v in k.items()}, "%.1f Gsamples/s" % (S * T * P * 2 / sum(k.values()) / 1e6))

tune = {k: int(v) for k, v in (kv.split("=") for kv in sys.argv[1].split(","))} if len(sys.argv) > 1 else None   # e.g. walk_fma=4
run(False, tune=tune); run(True, tune=tune)
