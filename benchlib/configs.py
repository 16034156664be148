"""One configuration of BASELINE.json measured the way the headline is: PCM resident in HBM, T-block run-ahead calls, wall
clock over asynchronous calls, per-kernel times of THIS run (events bound to the dispatches, fe_engine_set_profiling(2)),
parity of the first call against the float64 convolution — and its `configs` entry with both roofs (HBM bytes from the
committed PMC passes where they are of these very kernels; K2's issued flops against the FP32 vector peak)."""
import time

import numpy as np

from .formulas import HBM_PEAK_GBS, conv_f64, rms, tiled_bytes, valu_fractions, walk_flops
from .power import PowerWatch
from .profiles import load_traffic, profile_applies, traffic_key

ROLE_NAMES = {"forward": "K1 forward", "mac": "K2 mac", "inverse": "K3 inverse"}

# The other single-GPU configurations of BASELINE.json (parity-test shapes: tests/test_configs_gpu.py), measured the same
# way as the headline: PCM resident in HBM, T-block run-ahead calls, HIP events per kernel.
OTHER_CONFIGS = {
    "cfg1": dict(S=1, C=2, size=65536, populated=123, rate=44100, frames=2646000, gpu_ref=False,
                 what="one 44.1 kHz stereo file of 60 s (2 646 000 frames = 322 blocks + one of 8 176 frames) through the shape of "
                      "demo-filters/lowpass (a 123-tap FIR in a 65 536-frame impulse file: size 65 536, K = 8, every partition "
                      "populated as zita's impdata_create populates them; /root/reference/demo-filters/lowpass/filter-44100.conf, "
                      "README.md:358-361) — BASELINE.json configs[0], the reference's own CPU-runnable case: the CPU figures are the "
                      "point, the GPU rate of the same filter stands beside them"),
    "cfg2": dict(S=1, C=2, size=204800, populated=178193, rate=44100,
                 what="one 44.1 kHz stereo stream, SantaLucia-shaped filter (178 193 taps at delay 500 + a dirac, size 204 800: "
                      "K = 25, 22 populated; /root/reference/demo-filters/SantaLucia/filter-44100.conf:39-53)"),
    "cfg4": dict(S=1, C=8, size=524288, populated=None, rate=96000,
                 what="one 96 kHz 8-channel stream, 8 diagonal paths of 524 288 taps (K = 64)"),
    # not a BASELINE.json configuration: the headline's batch through a FULL filter matrix (a true-stereo reverb: four
    # /impulse/read paths, zita-config.cc:55-177) — twice K2's arithmetic on the same bytes, where K2 is arithmetic-bound
    "matrix": dict(S=64, C=2, size=262144, populated=None, rate=44100, full=True, cpu_leg=False, no_longer=True,
                   what="cfg3's batch (64 stereo streams, 262 144 taps, K = 32) through a full 2 x 2 filter matrix: four paths, "
                        "every output the sum of two convolutions"),
}

def profile_kernels(eng, run, steps):
    """Per-kernel times of THIS run, milliseconds per launch: (dispatch, event_to_event).  `dispatch`: a start and a stop
    event bound to each dispatch (fe_engine_set_profiling(2)) — the command processor's begin-to-end of that packet, what
    rocprofv3's kernel trace prints — with the rounds running back to back as in the timed loop.  `event_to_event`: events
    recorded between the launches (mode 1), which hold the launch boundary behind each kernel; kept beside it."""
    eng.reset_profile()
    eng.set_profiling(2)
    for _ in range(steps):
        run()
    eng.synchronize()
    prof = eng.get_kernel_profile()
    eng.set_profiling(1)
    for _ in range(min(steps, 60)):
        run()
    eng.synchronize()
    ev = eng.get_profile()
    eng.set_profiling(0)
    return ({k: v["ms"] / max(1, v["launches"]) for k, v in prof.items()},
            {k: v["ms"] / max(1, v["launches"]) for k, v in ev.items()})


def measure_config(S, C, size, T, populated=None, steps=100, warmup=10, tune=None, dev=0, check=True, frames=None, full=False, **_):
    """One filter of C diagonal paths (`populated` taps at offset 500 plus a dirac at 0, or `size` dense taps), S streams,
    T-block calls.  Returns ms per call (wall clock over `steps` asynchronous calls), per-kernel ms of this run
    (profile_kernels, a second loop), and — check=True — the rms deviation of the first call's output from the float64
    convolution."""
    import torch
    import folve_amd as fa
    from folve_amd.capi import BatchPlan, FE_DEVICE_PTRS, FE_ASYNC
    ts = torch.cuda.Stream()
    eng = fa.Engine(dev, ts.cuda_stream)
    if tune:
        eng.set_tuning(**tune)
    flt = fa.Filter(eng, C, C, size)
    rng = np.random.default_rng(3)
    taps = []
    cross = {}                                           # full matrix: taps of path (input, output)
    if full:
        for i in range(C):
            for o in range(C):
                h = rng.standard_normal(size).astype(np.float32)
                h *= np.float32(0.5) / np.linalg.norm(h)
                flt.add(i, o, h)
                cross[(i, o)] = h
    for c in range(0 if full else C):
        h = np.zeros(size, np.float32)
        if populated and populated < 4096:
            # a short FIR in a long impulse file (the lowpass demo): /impulse/read hands the engine the WHOLE file, zeros
            # included, and every partition the index range touches is populated
            ir = rng.standard_normal(populated).astype(np.float32)
            h[:populated] = ir / np.linalg.norm(ir)
            flt.add(c, c, h)
        elif populated:
            ir = (rng.standard_normal(populated) * np.exp(-np.arange(populated) / 40000.0)).astype(np.float32)
            h[500:500 + populated] = ir / np.linalg.norm(ir)
            h[0] += np.float32(0.4)
            flt.add(c, c, h[500:500 + populated], 500)
            flt.add(c, c, h[:1], 0)
        else:
            h = rng.standard_normal(size).astype(np.float32)
            h /= np.linalg.norm(h)
            flt.add(c, c, h)
        taps.append(h)
    flt.commit()
    P, K = flt.block_size, flt.partitions
    if frames:
        T = (frames + P - 1) // P                        # a whole file per call, its last block short
    nfr = frames or T * P
    streams = [flt.open_stream(T) for _ in range(S)]
    with torch.cuda.stream(ts):
        xs = [torch.rand(nfr, C, device="cuda") * 2 - 1 for _ in range(S)]
        ys = [torch.empty_like(x) for x in xs]
    plan = BatchPlan(streams, [x.data_ptr() for x in xs], [y.data_ptr() for y in ys], [nfr] * S, FE_DEVICE_PTRS | FE_ASYNC)
    parity = None
    if check:
        plan.run()
        eng.synchronize()
        torch.cuda.synchronize()
        n = min(T, 2 * K + 8) * P                        # long enough for every partition to act
        x0, y0 = xs[0][:n].cpu().numpy(), ys[0][:n].cpu().numpy()
        cc = [0, C - 1]
        if full:
            ref = sum(conv_f64(x0[:, [i] * len(cc)], [cross[(i, o)] for o in cc]) for i in range(C))
        else:
            ref = conv_f64(x0[:, cc], [taps[c] for c in cc])
        parity = max(rms(y0[:, cc] - ref), rms(y0[:, cc] - ref) / rms(ref))
        for st in streams:
            st.reset()
    for _ in range(warmup):
        plan.run()
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.run()
    eng.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # socket power and shader clock while these launches run: the same loop for at least 0.3 s under the reader (the timed
    # loop of a lone stream is over in 6 ms: no sample would fall into it)
    watch = PowerWatch(dev)
    with watch:
        for _ in range(max(steps, int(0.3 / max(dt, 1e-6)))):
            plan.run()
        eng.synchronize()
    kms, event_ms = profile_kernels(eng, plan.run, steps)
    launched = eng.last_kernels()
    out = {"kernels_launched": launched, "streams": S, "channels": C, "taps": size, "block": P, "partitions": K, "populated_partitions": flt.path_partitions(0, 0),
           "blocks_per_call": T, "ms_per_call": dt * 1e3, "kernels_ms": kms, "event_ms": event_ms, "msamples_per_s": S * nfr * C / dt / 1e6,
           "frames_per_call": nfr, "parity_rms": parity, "power": watch.summary(), "paths_per_output": C if full else 1}
    for s_ in streams:
        s_.close()
    del xs, ys
    return out

def measure_mixed_filters(dev=0, T=64, steps=60, warmup=8, sizes=(65536, 204800, 262144, 524288), per_filter=16):
    """Batches that mix filters: the reference resolves a configuration per sampling rate / channels / bits
    (/root/reference/processor-pool.cc:53-61), so a music library keeps several filters live and a combined batch holds
    streams of all of them.  64 stereo streams over 4 filters (K = 8 / 25 / 32 / 64) in ONE fe_batch_process call of
    T-block run-ahead chunks, against 64 streams of the one K = 32 filter in the same kind of call (about the same
    arithmetic: the mixed batch averages K = 32.25).  PCM resident in HBM; parity of one stream per filter against float64."""
    import torch
    import folve_amd as fa
    from folve_amd.capi import BatchPlan, FE_DEVICE_PTRS, FE_ASYNC
    ts = torch.cuda.Stream()
    eng = fa.Engine(dev, ts.cuda_stream)
    rng = np.random.default_rng(11)
    C = 2

    def make_filter(size):
        flt = fa.Filter(eng, C, C, size)
        taps = []
        for c in range(C):
            h = rng.standard_normal(size).astype(np.float32)
            h /= np.linalg.norm(h)
            flt.add(c, c, h)
            taps.append(h)
        flt.commit()
        return flt, taps

    def run(filters, counts):
        streams, taps_of = [], []
        for (flt, taps), n in zip(filters, counts):
            for _ in range(n):
                streams.append(flt.open_stream(T))
                taps_of.append(taps)
        # interleave the filters' streams, as open files arrive in any order
        order = sorted(range(len(streams)), key=lambda i: (i % per_filter, i // per_filter)) if len(filters) > 1 else list(range(len(streams)))
        streams = [streams[i] for i in order]
        taps_of = [taps_of[i] for i in order]
        P = filters[0][0].block_size
        with torch.cuda.stream(ts):
            xs = [torch.rand(T * P, C, device="cuda") * 2 - 1 for _ in streams]
            ys = [torch.empty_like(x) for x in xs]
        plan = BatchPlan(streams, [x.data_ptr() for x in xs], [y.data_ptr() for y in ys], [T * P] * len(streams), FE_DEVICE_PTRS | FE_ASYNC)
        plan.run()
        eng.synchronize()
        torch.cuda.synchronize()
        worst = 0.0
        for i in range(min(len(filters), len(streams))):         # the first stream of every filter (they are interleaved)
            n = min(T, 12) * P
            ref = conv_f64(xs[i][:n].cpu().numpy(), taps_of[i])
            worst = max(worst, rms(ys[i][:n].cpu().numpy() - ref))
        for _ in range(warmup):
            plan.run()
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            plan.run()
        eng.synchronize()
        dt = (time.perf_counter() - t0) / steps
        for s_ in streams:
            s_.close()
        return {"ms_per_call": round(dt * 1e3, 4), "msamples_per_s": round(len(streams) * T * P * C / dt / 1e6, 1), "parity_rms": worst}

    filters = [make_filter(sz) for sz in sizes]
    mixed = run(filters, [per_filter] * len(sizes))
    one = run([filters[2]], [per_filter * len(sizes)])
    return {"what": "%d stereo streams over %d filters of %s taps (K = %s) in one %d-block-per-stream call, against %d streams of "
                    "the %d-tap filter alone; PCM resident in HBM" % (per_filter * len(sizes), len(sizes), "/".join(str(z) for z in sizes),
                                                                      "/".join(str(f[0].partitions) for f in filters), T,
                                                                      per_filter * len(sizes), sizes[2]),
            "mixed": mixed, "one_filter": one, "mixed_over_one_filter": round(mixed["msamples_per_s"] / one["msamples_per_s"], 3)}

def kernel_table(kms, event_ms, launched, entry, by, tb, units):
    """Per kernel: this run's dispatch time (`ms`), the event-to-event time beside it, the committed profile's kernel-trace
    average for comparison (`trace_us`), counter bytes where the profile applies, and both HBM fractions."""
    trace = entry.get("avg_ns") or {}
    out = {}
    for k in kms:
        t = kms[k] * 1e-3
        out[k] = {"ms": round(kms[k], 4), "event_ms": round(event_ms[k], 4) if event_ms and k in event_ms else None,
                  "trace_us": round(trace[k] / 1e3, 2) if trace.get(k) else None,
                  "traffic": by.get(k),
                  "frac": round(by[k] / t / 1e9 / HBM_PEAK_GBS, 4) if by.get(k) and t else None,
                  "frac_of_min_bytes": round(tb[k] * units / t / 1e9 / HBM_PEAK_GBS, 4) if t else None,
                  "kernel": (launched or {}).get(k), "profiled_kernel": (entry.get("kernels") or {}).get(k)}
    return out


def choose_bound(hbm_frac, valu):
    """The roof that binds a launch: the larger of its HBM fraction and K2's FP32-vector fraction (nominal peak)."""
    vf = (valu or {}).get("frac")
    if vf is not None and (hbm_frac is None or vf > hbm_frac):
        return "valu"
    return "hbm"


def config_line(name, T, steps=100, tune=None, dev=0, check=True, cpu=False, longer_calls=True):
    """The `configs` entry of one configuration: rate at T-block calls, in-run per-kernel times, and its rooflines — HBM
    bytes per launch from the committed rocprofv3 PMC passes of `python bench.py --only-config <name>`
    (profiles/traffic.json), used only for the same kernels at times that agree; K2's issued flops against the FP32
    vector peak (nominal, and at the shader clock measured while the loop ran)."""
    cfg = OTHER_CONFIGS[name]
    r = measure_config(T=T, steps=steps, tune=tune, dev=dev, check=check, **cfg)
    P, K, C, S = r["block"], r["partitions"], r["channels"], r["streams"]
    T = r["blocks_per_call"]
    units = S * C * T
    tb = tiled_bytes(P, K, T)
    kms = r["kernels_ms"]
    entry = load_traffic().get(traffic_key(S, T, K, C, cfg.get("full"))) or {}
    by = entry.get("bytes") or {}
    launched = r["kernels_launched"]
    # (dispatch times against the profile's kernel-trace averages — the same kind of figure: 20 % or 3 us, and the same
    # kernels by name)
    ok, note = profile_applies(entry, launched, kms, 0.20, 3.0)
    if by and not ok:
        by = {}
    path_bytes = sum(by.values()) if len(by) == 3 else None
    kernels = kernel_table(kms, r.get("event_ms"), launched, entry, by, tb, units)
    dominant = max(kms, key=kms.get)
    sclk = (r.get("power") or {}).get("sclk_mhz")
    valu = valu_fractions(walk_flops(launched.get("mac"), P, K, T, S * C, r["paths_per_output"]), kms["mac"], sclk)
    hbm_frac = kernels[dominant]["frac"] if kernels[dominant]["frac"] is not None else kernels[dominant]["frac_of_min_bytes"]
    bound = choose_bound(hbm_frac, valu) if dominant == "mac" else "hbm"
    # The same stream in longer calls: a one-stream call is launch-chain bound (three dependent kernel boundaries whatever
    # the call's length), so the run-ahead depth the caller chooses sets how much of the roof a lone stream sees.
    # Reported beside the 256-block figure, never instead of it.
    longer = None
    if longer_calls and not cfg.get("frames") and not cfg.get("no_longer") and T < 1024:
        try:
            r4 = measure_config(T=1024, steps=max(20, steps // 3), tune=tune, dev=dev, check=False, **cfg)
            tb4 = tiled_bytes(P, K, 1024)
            longer = {"blocks_per_call": 1024, "ms_per_call": round(r4["ms_per_call"], 4), "msamples_per_s": round(r4["msamples_per_s"], 1),
                      "path_frac_of_min_bytes": round(tb4["total"] * S * C * 1024 / (r4["ms_per_call"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "kernels_ms": {k: round(v, 4) for k, v in r4["kernels_ms"].items()}}
        except Exception as e:  # noqa: BLE001
            longer = {"error": repr(e)}
    cpu_leg = None
    if cpu and cfg.get("cpu_leg", True):
        try:
            from .cpu import cpu_for_config
            cpu_leg = cpu_for_config(cfg)
            one = cpu_leg["one_stream_one_core"]["value"]
            cpu_leg["gpu_over_one_core"] = round(r["msamples_per_s"] / one, 1) if one else None
        except Exception as e:  # noqa: BLE001
            cpu_leg = {"error": repr(e)}
    why_none = None
    if not by:
        why_none = note or "no PMC traffic profiled for this shape (profiles/traffic.json has no entry %s)" % traffic_key(S, T, K, C, cfg.get("full"))
    wall = r["ms_per_call"] * 1e-3
    return {"workload": "%s: %s; P=%d, %d blocks per call, PCM resident in HBM" % (name, cfg["what"], P, T),
            "msamples_per_s": round(r["msamples_per_s"], 1), "ms_per_call": round(r["ms_per_call"], 4),
            "realtime_factor": round(r["frames_per_call"] / wall / cfg["rate"], 0),
            "cpu": cpu_leg, "longer_calls": longer,
            "blocks_per_call": T, "partitions": K, "populated_partitions": r["populated_partitions"],
            "parity_rms": r["parity_rms"], "kernels_launched": launched, "power": r.get("power"),
            "kernels_sum_ms": round(sum(kms.values()), 4),
            "roofline": {"bound": bound, "kernel": ROLE_NAMES[dominant], "kernel_name": launched.get(dominant),
                         "achieved": round(by[dominant] / (kms[dominant] * 1e-3) / 1e9, 1) if by.get(dominant) else None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": kernels[dominant]["frac"], "traffic": by.get(dominant),
                         "traffic_source": entry.get("profile"), "traffic_note": note,
                         "kernel_ms": round(kms[dominant], 4),
                         "time_source": "events bound to the dispatches in this run (fe_engine_set_profiling(2))",
                         # never "no roofline": without usable counter bytes the fraction by the MINIMUM bytes the call
                         # must move (every real kernel moves at least those) is a lower bound of the true fraction
                         "frac_lower_bound": kernels[dominant]["frac_of_min_bytes"],
                         "frac_lower_bound_why": why_none or "counter bytes are available: `frac` is the measured fraction, this its floor",
                         "k2_valu": valu,
                         "path": {"frac": round(path_bytes / wall / 1e9 / HBM_PEAK_GBS, 4) if path_bytes else None,
                                  "traffic": path_bytes,
                                  "frac_of_min_bytes": round(tb["total"] * units / wall / 1e9 / HBM_PEAK_GBS, 4),
                                  "min_bytes_per_call": int(tb["total"] * units)},
                         "kernels": kernels}}
