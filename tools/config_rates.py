"""Throughput and per-kernel times of the other BASELINE.json configurations (bench.py's `configs` leg uses the same
function).  Dev aid:  python tools/config_rates.py [blocks per call] [tune, e.g. mac_form=16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def measure(S, C, size, T, populated=None, dirac=None, steps=50, tune=None, dev=0):
    """One filter of C diagonal paths (`populated` taps at offset 500 + a dirac, or `size` dense taps), S streams,
    T-block calls with PCM resident in HBM.  Returns ms per call, per-kernel ms (HIP events), shape facts."""
    import torch
    import folve_amd as fa
    from folve_amd.capi import BatchPlan, FE_DEVICE_PTRS, FE_ASYNC
    ts = torch.cuda.Stream()
    eng = fa.Engine(dev, ts.cuda_stream)
    if tune:
        eng.set_tuning(**tune)
    flt = fa.Filter(eng, C, C, size)
    rng = np.random.default_rng(3)
    for c in range(C):
        if populated:
            h = (rng.standard_normal(populated) * np.exp(-np.arange(populated) / 40000.0)).astype(np.float32)
            flt.add(c, c, h / np.linalg.norm(h), 500)
            flt.add(c, c, np.float32([dirac or 0.4]), 0)
        else:
            h = rng.standard_normal(size).astype(np.float32)
            flt.add(c, c, h / np.linalg.norm(h))
    flt.commit()
    P, K = flt.block_size, flt.partitions
    streams = [flt.open_stream(T) for _ in range(S)]
    with torch.cuda.stream(ts):
        xs = [torch.rand(T * P, C, device="cuda") * 2 - 1 for _ in range(S)]
        ys = [torch.empty_like(x) for x in xs]
    plan = BatchPlan(streams, [x.data_ptr() for x in xs], [y.data_ptr() for y in ys], [T * P] * S, FE_DEVICE_PTRS | FE_ASYNC)
    for _ in range(5):
        plan.run()
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.run()
    eng.synchronize()
    dt = (time.perf_counter() - t0) / steps
    eng.set_profiling(True)
    eng.reset_profile()
    for _ in range(steps):
        plan.run()
    eng.synchronize()
    prof = eng.get_profile()
    eng.set_profiling(False)
    kms = {k: v["ms"] / max(1, v["launches"]) for k, v in prof.items()}
    out = {"streams": S, "channels": C, "taps": size, "block": P, "partitions": K, "populated_partitions": flt.path_partitions(0, 0),
           "blocks_per_call": T, "ms_per_call": dt * 1e3, "kernels_ms": kms, "msamples_per_s": S * T * P * C / dt / 1e6}
    for s_ in streams:
        s_.close()
    del xs, ys
    return out


if __name__ == "__main__":
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    tune = {k: int(v) for k, v in (kv.split("=") for kv in sys.argv[2].split(","))} if len(sys.argv) > 2 else None
    for name, kw in (("cfg2", dict(S=1, C=2, size=204800, populated=178193)),
                     ("cfg4", dict(S=1, C=8, size=524288)),
                     ("maxsize mono", dict(S=1, C=1, size=1048576)),
                     ("cfg3", dict(S=64, C=2, size=262144))):
        for t in ((T, 32) if name != "cfg3" else (T,)):
            r = measure(T=t, tune=tune, **kw)
            # bytes a T-block call must move at least (bench.py tiled_bytes): per block-channel 12P + 8P + 8P(T+K)/T + 8P + 12P... 
            P, K = r["block"], r["partitions"]
            minb = (4 * P + 8 * P + 8 * P * (t + K) / t + 8 * P + 8 * P + 4 * P) * r["channels"] * t * r["streams"]
            print("%-13s T=%3d K=%3d(%3d): %8.3f ms/call  K1 %.3f K2 %.3f K3 %.3f ms  %9.1f Msamples/s  min-bytes/t = %.2f TB/s"
                  % (name, t, K, r["populated_partitions"], r["ms_per_call"], r["kernels_ms"]["forward"], r["kernels_ms"]["mac"],
                     r["kernels_ms"]["inverse"], r["msamples_per_s"], minb / (r["ms_per_call"] * 1e-3) / 1e12))
