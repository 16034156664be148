import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import folve_amd as fa
from folve_amd.capi import BatchPlan, FE_DEVICE_PTRS, FE_ASYNC
from scipy.signal import fftconvolve

def run(name, S, C, size, frames, full=False, tune=None):
    ts = torch.cuda.Stream(); eng = fa.Engine(0, ts.cuda_stream)
    if tune: eng.set_tuning(**tune)
    flt = fa.Filter(eng, C, C, size); rng = np.random.default_rng(3)
    taps = {}
    for i in range(C):
        for o in range(C):
            if full or i == o:
                h = rng.standard_normal(size).astype(np.float32); h = h / np.linalg.norm(h) * 0.5; flt.add(i, o, h); taps[(i, o)] = h
    flt.commit(); P = flt.block_size; T = frames // P
    st = [flt.open_stream(T) for _ in range(S)]
    with torch.cuda.stream(ts):
        xs = [torch.rand(T * P, C, device="cuda") * 2 - 1 for _ in range(S)]; ys = [torch.empty_like(x) for x in xs]
    plan = BatchPlan(st, [x.data_ptr() for x in xs], [y.data_ptr() for y in ys], [T * P] * S, FE_DEVICE_PTRS | FE_ASYNC)
    plan.run(); eng.synchronize(); torch.cuda.synchronize()
    n = min(T * P, size + 3 * P)
    x0 = xs[0][:n].cpu().numpy().astype(np.float64); y0 = ys[0][:n].cpu().numpy()
    ref = np.zeros((n, C))
    for (i, o), h in taps.items():
        if o in (0, C - 1): ref[:, o] += fftconvolve(x0[:, i], h.astype(np.float64))[:n]
    err = max(np.sqrt(np.mean((y0[:, o] - ref[:, o]) ** 2)) for o in (0, C - 1))
    for s_ in st: s_.reset()
    for _ in range(10): plan.run()
    eng.synchronize(); t0 = time.perf_counter()
    for _ in range(60): plan.run()
    eng.synchronize(); dt = (time.perf_counter() - t0) / 60
    eng.set_profiling(True); eng.reset_profile()
    for _ in range(30): plan.run()
    eng.synchronize(); p = eng.get_profile(); eng.set_profiling(False)
    k = {n_: v["ms"] / v["launches"] for n_, v in p.items()}
    print("%-14s P=%5d T=%3d K=%3d  %.4f ms/call  %7.1f Gsamples/s  kernels %s  rms %.2e" % (name, P, T, flt.partitions, dt * 1e3, S * T * P * C / dt / 1e9, {n_: round(v, 4) for n_, v in k.items()}, err), flush=True)

which = sys.argv[1:] or ["cfg2", "cfg4", "cfg3", "matrix"]
if "cfg2" in which: run("cfg2-like", 1, 2, 204800, 256 * 8192)
if "cfg4" in which: run("cfg4", 1, 8, 524288, 256 * 8192)
if "cfg3" in which: run("cfg3", 64, 2, 262144, 256 * 8192)
if "matrix" in which: run("cfg3 2x2", 64, 2, 262144, 256 * 8192, full=True)
