"""Throughput of the other BASELINE.json configurations (parity-test shapes; bench.py measures cfg3). Dev aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import folve_amd as fa
from folve_amd.capi import BatchPlan, FE_DEVICE_PTRS, FE_ASYNC


def run(name, S, C, size, T, populated=None, steps=20):
    ts = torch.cuda.Stream()
    eng = fa.Engine(0, ts.cuda_stream)
    flt = fa.Filter(eng, C, C, size)
    rng = np.random.default_rng(3)
    n = populated or size
    for c in range(C):
        h = rng.standard_normal(n).astype(np.float32); h /= np.linalg.norm(h)
        flt.add(c, c, h)
    flt.commit()
    P, K = flt.block_size, flt.partitions
    streams = [flt.open_stream(T) for _ in range(S)]
    with torch.cuda.stream(ts):
        xs = [torch.rand(T * P, C, device="cuda") * 2 - 1 for _ in range(S)]
        ys = [torch.empty_like(x) for x in xs]
    plan = BatchPlan(streams, [x.data_ptr() for x in xs], [y.data_ptr() for y in ys], [T * P] * S, FE_DEVICE_PTRS | FE_ASYNC)
    for _ in range(3):
        plan.run()
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.run()
    eng.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("%s: S=%d C=%d K=%d(%d populated) T=%d: %.3f ms/step, %.1f Msamples/s, %.0fx real time per stream" % (
        name, S, C, K, flt.path_partitions(0, 0), T, dt * 1e3, S * T * P * C / dt / 1e6,
        T * P / dt / (96000 if C == 8 else 44100)))


run("cfg2", 1, 2, 204800, 32, populated=178693)
run("cfg2 one block per call", 1, 2, 204800, 1, populated=178693, steps=100)
run("cfg4", 1, 8, 524288, 32)
run("cfg4 one block per call", 1, 8, 524288, 1, steps=100)
run("cfg3", 64, 2, 262144, 32)
