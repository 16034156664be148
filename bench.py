#!/usr/bin/env python3
"""bench.py — throughput of the folve convolution hot path on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
64 concurrent synthetic 44.1 kHz / 2-channel streams per GPU through one shared
2-path 262 144-tap random FIR (P = 8192, K = 32), PCM resident in HBM.  One
"step" = one batched pass of the hot path (K1 forward FFT -> K2 MAC -> K3
inverse FFT) over `--blocks` consecutive 8192-frame blocks of every stream
(run-ahead batches, the BufferThread role in folve; default 256 = the >= 256 blocks
cfg3 gives each stream, i.e. one step is one pass over the whole synthetic input).  N > 1 shards independent
streams across GPUs (64 per GPU, cfg5 = 512 streams on 8): no data-path
collective, torch.distributed (RCCL) only for the barrier and the max-over-ranks.

Prints ONE JSON line on rank 0 (contract in the task statement).  Beside the
contract's keys:
  parity_rms   — gate, BEFORE anything is timed: the first two steps' output of two streams
                 against the float64 linear convolution; exit 1 above 1e-5
  roofline     — the dominant kernel against the HBM roof.  `frac` = HBM bytes the kernel really
                 moves (rocprofv3 PMC, profiles/traffic.json, taken from the SAME command) / its
                 HIP-event time / 8 TB/s: always <= 1.  The streaming formula of SURVEY.md §8(d)
                 does not describe a time-tiled kernel (it re-uses rows on chip); it is printed as
                 `frac_alg` with "applicable": false, and applies in `roofline_streaming`
                 `roofline.measured_hbm`: this GPU's read / write / copy rates measured in this run by plain
                 streaming kernels (fe_engine_hbm_rates), and every kernel's time against its own PMC bytes
                 at those rates — what "speed of light" is for a kernel that writes
  steady_state — 400 more steps of the same launches, with socket power and shader clock (amdgpu hwmon)
  roofline_streaming — one block per call (SoundProcessor::Process granularity): K2 streams K rows
                 per block, algorithmic and moved bytes coincide
  end_to_end   — the same batch from page-locked HOST buffers, PCIe inside the timed region
  single_block_us — one synchronous stereo block through fe_stream_process (the drop-in call)
  drop_in_threads — the same call from 1 / 16 / 64 host threads at once, each its own folve::SoundProcessor
                 (a C++ child process over include/folve_host.h), with and without the per-GPU combiner
  configs      — cfg1 (the lowpass demo shape, a 60 s file), cfg2 (one stereo stream, SantaLucia shape) and cfg4 (96 kHz x
                 8 channels x 512 k taps) at 256-block calls: rate, per-kernel ms and roofline (PMC bytes of
                 `bench.py --only-config cfgN`, profiles/traffic.json), each with the CPU path on the same shape (`cpu`)
  mixed_filters — 64 streams over 4 filters (K = 8/25/32/64) in one call against 64 streams of one filter
  cpu_baseline — the CPU restatement of zita-convolver's algorithm (oracle/, rebuilt -march=native
                 on this box), all cores and one core, on a bounded sample (N = 1 only)
"""
import argparse
import ctypes
import ctypes.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FS = 44100
PARITY_TOL = 1e-5


def alg_bytes(P, K_eff, S):
    """bytes per block-channel of the STREAMING uniformly partitioned algorithm
    (every output block reads K spectra of its stream and K of the filter), and its per-kernel split."""
    fwd = 4 * P + 8 * (P + 1)                 # read the block's PCM once; write one spectrum
    mac = 8 * (P + 1) * K_eff + 8 * (P + 1) * K_eff / S   # read K spectra + the shared filter
    inv = 8 * (P + 1) + 4 * P                 # read the accumulated spectrum; write P samples
    return {"forward": fwd, "mac": mac, "inverse": inv, "total": 12 * P + 8 * (P + 1) * (K_eff + 1) + 8 * (P + 1) * K_eff / S}


def tiled_bytes(P, K, T):
    """Bytes per block-channel a run-ahead call of T blocks must move at least: every PCM sample in
    and out once, every spectrum written once and read once by K2 (plus the K history rows per call),
    every accumulated spectrum written and read once."""
    fwd = 4 * P + 8 * P
    mac = 8 * P * (T + K) / T + 8 * P
    inv = 8 * P + 4 * P
    return {"forward": fwd, "mac": mac, "inverse": inv, "total": fwd + mac + inv}


def conv_f64(x, taps):
    """Exact causal linear convolution per channel, float64, truncated to len(x) (the ground truth)."""
    from scipy.signal import fftconvolve
    n = x.shape[0]
    return np.stack([fftconvolve(x[:, c].astype(np.float64), taps[c].astype(np.float64))[:n] for c in range(x.shape[1])], 1)


class PowerWatch:
    """Socket power and shader clock of one GPU while a loop runs: a thread reading the amdgpu hwmon files
    (power1_input, power1_cap, freq1_input).  The device is found by PCI address; failing that, the
    busiest amdgpu card at sampling time.  Informational: says whether a step runs against the power cap."""

    def __init__(self, device):
        import glob
        self.dirs = []
        try:
            pr = torch.cuda.get_device_properties(device)
            pat = "/sys/bus/pci/devices/%04x:%02x:%02x.0/hwmon/hwmon*" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            self.dirs = glob.glob(pat)
        except Exception:
            self.dirs = []
        self.by_address = bool(self.dirs)
        if not self.dirs:
            self.dirs = [d for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
                         if os.path.exists(os.path.join(d, "power1_input"))]
        self.samples = []
        self._stop = None
        self._thread = None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return int(f.read().strip())
        except Exception:
            return None

    def _sample(self):
        best = None
        for d in self.dirs:
            pw = self._read(os.path.join(d, "power1_input"))
            if pw is None:
                pw = self._read(os.path.join(d, "power1_average"))
            if pw is not None and (best is None or pw > best[0]):
                best = (pw, self._read(os.path.join(d, "freq1_input")), self._read(os.path.join(d, "power1_cap")), d)
        if best:
            self.samples.append(best)

    def __enter__(self):
        import threading
        self._stop = threading.Event()

        def run():
            while not self._stop.is_set():
                self._sample()
                self._stop.wait(0.02)
        if self.dirs:
            self._thread = threading.Thread(target=run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        if self._thread:
            self._stop.set()
            self._thread.join()

    def summary(self):
        if len(self.samples) < 3:
            return None
        mid = self.samples[len(self.samples) // 4:]              # the controller needs a moment to react
        pw = sorted(x[0] for x in mid)[len(mid) // 2] / 1e6
        fr = sorted(x[1] or 0 for x in mid)[len(mid) // 2] / 1e6
        cap = (mid[-1][2] or 0) / 1e6
        top = None
        try:
            with open(os.path.join(os.path.dirname(os.path.dirname(mid[-1][3])), "pp_dpm_sclk")) as f:
                top = max(int(l.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", "")) for l in f if ":" in l)
        except Exception:
            top = None
        return {"socket_w": round(pw, 1), "cap_w": round(cap, 1), "sclk_mhz": round(fr, 1), "sclk_max_mhz": top,
                "at_power_cap": bool(cap and pw >= 0.97 * cap), "samples": len(mid),
                "source": "amdgpu hwmon (power1_input, freq1_input), device by " + ("PCI address" if self.by_address else "highest power")}


def norm_kernel(name):
    """'mac_walk_kernel<33, 7, true, 4, 1, 1> grid=1048576' (profiles/traffic.json) or the engine's own spelling
    (fe_engine_last_kernels) -> 'mac_walk_kernel<33,7,true,4,1,1>'."""
    return (name or "").split(" grid=")[0].replace(" ", "")


def profile_applies(entry, launched, kms, rel, floor_us, roles=None):
    """Is profiles/traffic.json's `entry` a profile of THIS run's launches?  Its kernels must be the ones the engine
    launched — by NAME (the instantiation, as rocprofv3 and fe_engine_last_kernels both spell it) — and this run's kernel
    times must agree with the profile's kernel-trace averages within `rel` (or `floor_us`: HIP events around a short launch
    read a few microseconds long).  Returns (ok, note)."""
    by = entry.get("bytes") or {}
    if not by:
        return False, None
    for k in (roles or by):
        prof, ran = norm_kernel((entry.get("kernels") or {}).get(k)), norm_kernel((launched or {}).get(k))
        if not prof or not ran or prof != ran:
            return False, ("profile %s is of other kernels (%s: profiled %s, launched %s): its traffic is not used"
                           % (entry.get("profile"), k, prof or "?", ran or "?"))
    for k in (roles or by):
        ns = (entry.get("avg_ns") or {}).get(k, 0)
        if abs(kms[k] * 1e6 - ns) > max(rel * ns, floor_us * 1e3):
            return False, ("in-run %s time %.1f us differs from profile %s's %.1f us by more than %.0f %%: its traffic is not used"
                           % (k, kms[k] * 1e3, entry.get("profile"), ns / 1e3, rel * 100))
    return True, None


def usable_cpus():
    """(cpus this process may run on at once, host cpus, why): the affinity mask cut to the cgroup's CPU quota —
    256 threads inside a 16-CPU quota are 16 cores' worth of work with a throttle on top."""
    host = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = host
    why = "affinity mask"
    try:
        quota = None
        if os.path.exists("/sys/fs/cgroup/cpu.max"):                      # cgroup v2
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
        elif os.path.exists("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):      # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        if quota is not None and quota < n:
            n, why = max(1, int(quota)), "cgroup CPU quota"
    except Exception:
        pass
    return n, host, why


def rms(a):
    a = np.asarray(a, np.float64)
    return float(np.sqrt(np.mean(a * a)))


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


def fit_trace_to_wall(trace_ms, event_ms, wall_ms):
    """The factor (<= 1) by which a profile's kernel-trace times of a call's three kernels must shrink so that they do not
    exceed the call's wall time of THIS run."""
    if not trace_ms or len(trace_ms) != len(event_ms):       # (event times hold the launch boundaries: the two kinds do not add up)
        return 1.0
    tsum = sum(trace_ms.values())
    return 1.0 if (tsum <= wall_ms or tsum <= 0) else wall_ms / tsum


def traffic_key(S, T, K, C, full=False):
    return "S%d_T%d_K%d_C%d" % (S, T, K, C) + ("_full" if full else "")


def measure_config(S, C, size, T, populated=None, steps=100, warmup=10, tune=None, dev=0, check=True, frames=None, full=False, **_):
    """One filter of C diagonal paths (`populated` taps at offset 500 plus a dirac at 0, or `size` dense taps), S streams,
    T-block calls.  Returns ms per call (wall clock over `steps` asynchronous calls), per-kernel ms (HIP events, a second
    loop), and — check=True — the rms deviation of the first call's output from the float64 convolution."""
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
    eng.set_profiling(True)
    eng.reset_profile()
    for _ in range(steps):
        plan.run()
    eng.synchronize()
    prof = eng.get_profile()
    eng.set_profiling(False)
    kms = {k: v["ms"] / max(1, v["launches"]) for k, v in prof.items()}
    launched = eng.last_kernels()
    out = {"kernels_launched": launched, "streams": S, "channels": C, "taps": size, "block": P, "partitions": K, "populated_partitions": flt.path_partitions(0, 0),
           "blocks_per_call": T, "ms_per_call": dt * 1e3, "kernels_ms": kms, "msamples_per_s": S * nfr * C / dt / 1e6,
           "frames_per_call": nfr, "parity_rms": parity}
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


def cpu_for_config(cfg, budget=2.0):
    """The CPU path on a configuration's shape (the CPU beside the GPU, same shape): the vectorised
    stand-in (oracle/fastcpu.c) and the scalar parity oracle, ONE stream on one core — a configuration with one stream IS
    one thread in folve's threading model (one synchronous engine per open file) — and, for context, as many such streams
    as this process has CPUs.  Bounded: about `budget` seconds per configuration."""
    from oracle import oracle as O      # the reported baseline, not the product
    native = O.native_bench_lib() is not None
    C, size = cfg["C"], cfg["size"]
    P = O.fragm_for_size(size)
    cores = usable_cpus()[0]
    tp = O.fast_bench_streams(1, 4, 1, C, size, 3, native=native) / 4.0
    nb = int(max(8, min(cfg.get("frames", 10 ** 9) // P + 1 if cfg.get("frames") else 4096, 0.35 * budget / max(tp, 1e-6))))
    t1 = O.fast_bench_streams(1, nb, 1, C, size, 3, native=native)
    nba = int(max(8, min(nb, 0.35 * budget / max(tp * 2.5, 1e-6))))
    ta = O.fast_bench_streams(cores, nba, cores, C, size, 3, native=native)
    nbs = int(max(4, min(nb, 0.3 * budget / max(tp * 3.0, 1e-6))))
    ts_ = O.bench_streams(1, nbs, 1, C, C, size, 3, native=native)
    return {"kind": "port", "what": "oracle/fastcpu.c (vectorised stand-in for zita-convolver, which is unavailable offline) on this "
                                    "configuration's shape: %d channels, %d taps, partition %d, dense filter" % (C, size, P),
            "one_stream_one_core": {"value": round(nb * P * C / t1 / 1e6, 2), "unit": "Msamples/s", "cores": 1,
                                    "sample": "%d blocks, %.2f s" % (nb, t1),
                                    "realtime_factor": round(nb * P / t1 / cfg["rate"], 1)},
            "streams_on_all_cores": {"value": round(cores * nba * P * C / ta / 1e6, 2), "unit": "Msamples/s", "cores": cores,
                                     "sample": "%d such streams x %d blocks, %d threads, %.2f s" % (cores, nba, cores, ta)},
            "scalar_oracle_one_core": {"value": round(nbs * P * C / ts_ / 1e6, 2), "unit": "Msamples/s", "cores": 1,
                                       "sample": "%d blocks, %.2f s" % (nbs, ts_)},
            "build": "-O3 -march=native on this box" if native else "-O3 -march=x86-64-v3 (prebuilt)"}


def config_line(name, T, steps=100, tune=None, dev=0, check=True, cpu=False, longer_calls=True):
    """The `configs` entry of one configuration: rate at T-block calls, per-kernel times, and its roofline — HBM bytes per
    launch from the committed rocprofv3 PMC passes of `python bench.py --only-config <name>` (profiles/traffic.json), used
    only while this run's kernel times agree with the profiled run's."""
    cfg = OTHER_CONFIGS[name]
    r = measure_config(T=T, steps=steps, tune=tune, dev=dev, check=check, **cfg)
    P, K, C, S = r["block"], r["partitions"], r["channels"], r["streams"]
    T = r["blocks_per_call"]
    units = S * C * T
    tb = tiled_bytes(P, K, T)
    kms = r["kernels_ms"]
    dominant = max(kms, key=kms.get)
    entry = {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            entry = json.load(open(tpath)).get(traffic_key(S, T, K, C, cfg.get("full"))) or {}
        except Exception:
            entry = {}
    by = entry.get("bytes") or {}
    launched = r["kernels_launched"]
    # (HIP events around a kernel of a few tens of microseconds read 4 - 10 us long, more on a box whose clocks have not
    # settled: agreement within 25 % or 14 us — and the same kernels, by name)
    ok, note = profile_applies(entry, launched, kms, 0.25, 14.0)
    if by and not ok:
        by = {}
    path_bytes = sum(by.values()) if len(by) == 3 else None
    # A launch of up to ~150 us: the engine's HIP events stand one dependent-launch boundary (3 - 7 us) apart, so an
    # event-to-event time holds the kernel AND the gap behind it (the three of them can add up to more than the call's wall
    # time).  Where the committed profile is of these very kernels, such a launch's duration is the profile's kernel-trace
    # average (`trace_us`), printed beside the event time, and `frac` divides by that.
    trace = entry.get("avg_ns") or {}
    kernels = {}
    # (a box faster than the one that took the profile: the profile's kernel times, which this run's call cannot exceed, are
    # scaled down to the call's wall time — and say so)
    fit = fit_trace_to_wall({k: trace[k] / 1e6 for k in kms if by.get(k) and trace.get(k) and kms[k] < 0.15}, kms, r["ms_per_call"])
    for k in kms:
        short = by.get(k) and trace.get(k) and kms[k] < 0.15
        t_ms = trace[k] / 1e6 * fit if short else kms[k]
        kernels[k] = {"ms": round(t_ms, 4), "event_ms": round(kms[k], 4),
                      "trace_us": round(trace[k] / 1e3, 2) if (by.get(k) and trace.get(k)) else None,
                      "time_source": ("rocprofv3 kernel trace of profile %s (a launch this short: the event time includes the launch boundary)%s"
                                      % (entry.get("profile"), "" if fit == 1.0 else "; x %.3f: this box's call is shorter than the profiled kernels' sum" % fit))
                      if short else "HIP events in this run",
                      "traffic": by.get(k),
                      "frac": round(by[k] / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if by.get(k) else None,
                      "frac_of_min_bytes": round(tb[k] * units / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "kernel": launched.get(k), "profiled_kernel": (entry.get("kernels") or {}).get(k)}
    dominant = max(kernels, key=lambda k: kernels[k]["ms"])
    # The same stream in longer calls: a one-stream call is launch-chain bound (three dependent kernel boundaries of ~5.8 us
    # whatever the call's length), so the run-ahead depth the caller chooses sets how much of the
    # roof a lone stream sees.  Reported beside the 256-block figure, never instead of it.
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
            cpu_leg = cpu_for_config(cfg)
            one = cpu_leg["one_stream_one_core"]["value"]
            cpu_leg["gpu_over_one_core"] = round(r["msamples_per_s"] / one, 1) if one else None
        except Exception as e:  # noqa: BLE001
            cpu_leg = {"error": repr(e)}
    why_none = None
    if not by:
        why_none = note or "no PMC traffic profiled for this shape (profiles/traffic.json has no entry %s)" % traffic_key(S, T, K, C, cfg.get("full"))
    return {"workload": "%s: %s; P=%d, %d blocks per call, PCM resident in HBM" % (name, cfg["what"], P, T),
            "msamples_per_s": round(r["msamples_per_s"], 1), "ms_per_call": round(r["ms_per_call"], 4),
            "realtime_factor": round(r["frames_per_call"] / (r["ms_per_call"] * 1e-3) / cfg["rate"], 0),
            "cpu": cpu_leg, "longer_calls": longer,
            "blocks_per_call": T, "partitions": K, "populated_partitions": r["populated_partitions"],
            "parity_rms": r["parity_rms"], "kernels_launched": launched,
            "roofline": {"bound": "hbm", "kernel": {"forward": "K1 forward", "mac": "K2 mac", "inverse": "K3 inverse"}[dominant],
                         "achieved": round(by[dominant] / (kernels[dominant]["ms"] * 1e-3) / 1e9, 1) if by.get(dominant) else None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": kernels[dominant]["frac"], "traffic": by.get(dominant),
                         "traffic_source": entry.get("profile"), "traffic_note": note,
                         # never "no roofline": without usable counter bytes the fraction by the MINIMUM bytes the call
                         # must move (every real kernel moves at least those) is a lower bound of the true fraction
                         "frac_lower_bound": kernels[dominant]["frac_of_min_bytes"],
                         "frac_lower_bound_why": why_none or "counter bytes are available: `frac` is the measured fraction, this its floor",
                         "path": {"frac": round(path_bytes / (r["ms_per_call"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if path_bytes else None,
                                  "traffic": path_bytes,
                                  "frac_of_min_bytes": round(tb["total"] * units / (r["ms_per_call"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  "min_bytes_per_call": int(tb["total"] * units)},
                         "kernels": kernels}}


def cpu_baseline_leg(args, P, C, size):
    """The CPU path timed on this box's host cores, on a bounded sample of the benchmarked workload (rank 0 only; at
    N > 1 after the process group is gone, so that no rank waits in an RCCL barrier while the CPU works)."""
    from oracle import oracle as O      # CPU restatement: the baseline being reported, not the product
    native = O.native_bench_lib() is not None
    cores, host_cpus, cores_why = usable_cpus()
    # The real libzita-convolver, where this box has it ("additionally time the real thing through the same
    # harness"): tests/compile/zita_ref.cpp is built against it and runs the same shape — one Convproc per stream, configured as
    # folve configures it, streams dealt to threads — all cores and one core.  Absent (both boxes seen so far): says why.
    zita = {"available": False}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import zita_real
        zexe, zwhy = zita_real.build()
        if zexe is None:
            zita["why"] = zwhy
        else:
            zb = max(8, int(0.3 * args.cpu_seconds / max(1e-4, zita_real.bench(zexe, C, size, cores, 8, cores)["seconds"] / 8.0)))
            ra = zita_real.bench(zexe, C, size, cores, zb, cores)
            r1 = zita_real.bench(zexe, C, size, 1, max(16, zb), 1)
            zita = {"available": True, "kind": "reference", "unit": "Msamples/s", "zita_major": ra.get("zita_major"),
                    "value": round(cores * zb * P * C / ra["seconds"] / 1e6, 2), "cores": cores,
                    "sample": "%d streams x %d blocks x %d ch, %d taps, one Convproc per stream, %d threads, %.1f s" % (cores, zb, C, size, cores, ra["seconds"]),
                    "one_core": {"value": round(max(16, zb) * P * C / r1["seconds"] / 1e6, 2), "sample": "1 stream x %d blocks, %.1f s" % (max(16, zb), r1["seconds"])},
                    "what": "libzita-convolver itself through tests/compile/zita_ref.cpp (Convproc configured as /root/reference/zita-fconfig.cc:74-94, "
                            "blocks as sound-processor.cc:98-127)"}
    except Exception as ex:  # noqa: BLE001 - a reported extra: never fails the line
        zita = {"available": False, "why": repr(ex)}

    def timed(fn, budget):
        """(all-core rate, sample text, one-core rate, sample text) of one CPU engine, sized to `budget` seconds."""
        # a short probe runs ~2.5x faster per block than the steady state (cold DRAM working set of
        # 8 MB per stream builds up), so size the sample from a 16-block probe
        tprobe = fn(cores, 16, cores) / 16.0                                  # seconds per block round
        nblocks = int(max(8, min(65536, budget / max(tprobe * 1.5, 1e-4))))
        tall = fn(cores, nblocks, cores)
        tp1 = fn(1, 64, 1) / 64.0                                             # one stream alone is cache-resident: its own probe
        nb1 = int(max(64, min(65536, 0.4 * budget / max(tp1, 1e-6))))
        t1 = fn(1, nb1, 1)
        return (cores * nblocks * P * C / tall / 1e6,
                "%d streams x %d blocks x %d ch, %d taps, one convolver per stream, %d threads, %.1f s" % (cores, nblocks, C, size, cores, tall),
                nb1 * P * C / t1 / 1e6,
                "1 stream x %d blocks, 1 thread, %.1f s (one stream's 8 MB of state stays in cache)" % (nb1, t1))

    # the vectorised stand-in (oracle/fastcpu.c: split-complex radix-4 Stockham FFT, FMA multiply-accumulate) is the
    # figure to compare with; the scalar parity oracle is timed beside it
    fv, fs, f1, f1s = timed(lambda ns, nb, nt: O.fast_bench_streams(ns, nb, nt, C, size, 3, native=native), 0.6 * args.cpu_seconds)
    sv, ss, s1, s1s = timed(lambda ns, nb, nt: O.bench_streams(ns, nb, nt, C, C, size, 3, native=native), 0.4 * args.cpu_seconds)
    cpu = {"value": round(fv, 2), "unit": "Msamples/s", "cores": cores,
           "cores_note": "%d threads = the CPUs this process may use (%s); the host has %d" % (cores, cores_why, host_cpus),
           "kind": "port",
           "what": "CPU restatement of zita-convolver's algorithm as folve configures it (one level, partition 8192, one engine "
                   "per open file: /root/reference/zita-fconfig.cc:74-81), vectorised: split-complex radix-4 Stockham real FFT and "
                   "an FMA multiply-accumulate over structure-of-arrays spectra (oracle/fastcpu.c).  zita-convolver / FFTW are "
                   "unavailable offline: this is a stand-in, not zita.  Its time is the multiply-accumulate streaming K spectra "
                   "of the stream and of the filter per block (4 MB per channel and block) through the cache hierarchy.",
           "build": "-O3 -march=native on this box" if native else "-O3 -march=x86-64-v3 (prebuilt)",
           "sample": fs,
           "one_core": {"value": round(f1, 2), "unit": "Msamples/s", "cores": 1, "sample": f1s},
           "scalar_oracle": {"value": round(sv, 2), "unit": "Msamples/s", "cores": cores, "sample": ss,
                             "one_core": {"value": round(s1, 2), "sample": s1s},
                             "what": "the parity oracle itself (oracle_convproc.c + oracle_fft.c: scalar radix-2 FFT, written to be "
                                     "read): a pessimistic figure, kept for continuity with rounds 1 - 2"},
           "zita_convolver_on_this_box": zita}
    return cpu


def self_launch(n):
    """`python bench.py --gpus N` (N > 1) with no launcher around it: start the N ranks ourselves, one per GPU, as a CHILD
    process (`python -m torch.distributed.run`, rendezvous on 127.0.0.1 at a free port) and hand back its exit code.  This
    process has imported neither torch nor the engine at this point, so nothing here has touched the GPU; the child's
    stdout is ours, so rank 0's JSON line comes out unchanged."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: --gpus %d without a launcher: starting %d ranks (torch.distributed.run, port %d)\n" % (n, n, port))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--streams", type=int, default=64, help="streams per GPU")
    ap.add_argument("--blocks", type=int, default=256,
                    help="consecutive blocks per stream per step (run-ahead depth; cfg3 gives every stream >= 256 blocks: one step is the whole of it)")
    ap.add_argument("--taps", type=int, default=262144)
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the streaming / end-to-end / single-block legs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU baseline sample length (all-core leg)")
    ap.add_argument("--tune", default="", help="engine tuning for experiments, e.g. mac_form=16,fwd_run=8")
    ap.add_argument("--skip", default="", help="comma-separated extra legs to leave out: streaming,end_to_end,single_block,drop_in,configs,mixed")
    ap.add_argument("--only-config", default="", choices=["", "cfg1", "cfg2", "cfg4", "matrix"],
                    help="run only this configuration's loop and print its `configs` entry (what tools/profile.sh profiles)")
    ap.add_argument("--config-blocks", type=int, default=256, help="blocks per call of the cfg2 / cfg4 legs")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    if args.only_config:
        import torch  # noqa: F401
        tune = {k: int(v) for k, v in (kv.split("=") for kv in args.tune.split(","))} if args.tune else None
        # (--skip longer: without the 1 024-block leg, whose launches can share a kernel and a grid with the 256-block ones —
        # a profile of this command must not average the two)
        print(json.dumps({args.only_config: config_line(args.only_config, args.config_blocks, steps=min(args.steps, 300), tune=tune,
                                                        longer_calls="longer" not in args.skip.split(","))}))
        return

    import torch
    import folve_amd as fa
    from folve_amd.capi import BatchPlan, FE_ASYNC, FE_DEVICE_PTRS, FE_HOST_PTRS

    from folve_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # FOLVE_BENCH_DEVICE / FOLVE_BENCH_BACKEND exist so that the N > 1 code path can be exercised on a
    # one-GPU box (all ranks on one device, gloo); the driver's runs use one rank per GPU over RCCL.
    dev = int(os.environ.get("FOLVE_BENCH_DEVICE", local_rank if world > 1 else 0))
    backend = os.environ.get("FOLVE_BENCH_BACKEND", "nccl")
    if dev >= torch.cuda.device_count():
        # (one rank per GPU: `--gpus N` needs N visible devices — or FOLVE_BENCH_DEVICE to put every rank on one, as the tests do)
        sys.stderr.write("bench.py: rank %d wants GPU %d but only %d are visible (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?)\n"
                         % (rank, dev, torch.cuda.device_count()))
        if rank == 0:
            print(json.dumps({"error": "rank %d wants GPU %d, %d visible" % (rank, dev, torch.cuda.device_count()), "n_gpus": world}))
        sys.exit(2)
    torch.cuda.set_device(dev)
    dist = None
    # FOLVE_BENCH_FORCE_DIST=1: a process group even at world size 1, so that the RCCL branch below (init, barrier, the
    # reductions of sharding.aggregate_throughput beside the engine's own HIP streams) runs on a one-GPU box
    force_dist = world == 1 and os.environ.get("FOLVE_BENCH_FORCE_DIST", "") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dist and "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert args.gpus == world, "--gpus must equal WORLD_SIZE (launch N>1 with torch.distributed.run)"
    red_dev = torch.device("cuda", dev) if backend == "nccl" else torch.device("cpu")
    # streams are sharded by index, as the pool hands out processors: gpu = stream % world
    my_streams = sharding.shard_streams(args.streams * world, world, rank)
    assert len(my_streams) == args.streams

    S, T, C, size = args.streams, args.blocks, args.channels, args.taps
    ts = torch.cuda.Stream()
    eng = fa.Engine(dev, ts.cuda_stream)
    if args.tune:
        eng.set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in args.tune.split(","))})
    flt = fa.Filter(eng, C, C, size)
    P, K = flt.block_size, flt.partitions
    rng = np.random.default_rng(3)
    taps = []
    for c in range(C):                               # one shared filter, C diagonal paths, unit L2 norm
        h = rng.standard_normal(size).astype(np.float32)
        h /= np.linalg.norm(h)
        taps.append(h)
        flt.add(c, c, h)
    flt.commit()
    streams = [flt.open_stream(T) for _ in range(S)]
    with torch.cuda.stream(ts):
        xs, ys = [], []
        for s in range(S):
            g = torch.Generator(device="cuda")
            g.manual_seed(100 + my_streams[s])
            xs.append(torch.rand(T * P, C, device="cuda", generator=g) * 2 - 1)   # U(-1, 1)
            ys.append(torch.empty(T * P, C, device="cuda"))
    plan = BatchPlan(streams, [x.data_ptr() for x in xs], [y.data_ptr() for y in ys], [T * P] * S,
                     FE_DEVICE_PTRS | FE_ASYNC)

    def sync():
        eng.synchronize()
        torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    # ---- parity gate: nothing is timed unless the benchmarked launch is right ----
    # Step 1 runs from zeroed state, step 2 carries it: outputs of two streams against the float64
    # linear convolution of [x | x] with the taps.  The references are computed FIRST (seconds of CPU
    # work), so that the GPU does not sit idle between the gate's two steps and the warm-up.
    check = sorted({0, S - 1})
    sync()
    refs = {}
    for s in check:
        x = xs[s].cpu().numpy()
        refs[s] = conv_f64(np.concatenate([x, x]), taps)
    plan.run(); sync()
    y1 = {s: ys[s].cpu().numpy().copy() for s in check}
    plan.run(); sync()
    y2 = {s: ys[s].cpu().numpy().copy() for s in check}
    parity_abs, parity_rel = 0.0, 0.0
    for s in check:
        got = np.concatenate([y1[s], y2[s]])
        e = rms(got - refs[s])
        parity_abs = max(parity_abs, e)
        parity_rel = max(parity_rel, e / rms(refs[s]))
    del refs
    if dist is not None:
        t = torch.tensor([parity_abs, parity_rel], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        parity_abs, parity_rel = float(t[0]), float(t[1])
    if not (parity_abs <= PARITY_TOL and parity_rel <= PARITY_TOL):
        if rank == 0:
            print(json.dumps({"error": "parity gate failed", "parity_rms": parity_abs, "parity_rel": parity_rel,
                              "tolerance": PARITY_TOL}))
        sys.stderr.write("bench.py: PARITY GATE FAILED (rms %.3e, rel %.3e > %.0e): nothing was timed\n"
                         % (parity_abs, parity_rel, PARITY_TOL))
        sys.exit(1)

    for _ in range(args.warmup):
        plan.run()
    sync(); barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.run()
    sync(); barrier(); sync()
    dt = time.perf_counter() - t0
    _, dt, _ = sharding.aggregate_throughput(S * T * P * args.steps, dt, dist, red_dev)   # max over ranks
    # The hot path's one metric, max_output_value() (/root/reference/sound-processor.cc:116-125), of every stream of the job in
    # global stream order: each rank's K3 keeps its streams' running maxima on the GPU; the ranks' shards are disjoint, so one
    # sum-reduction of a 64 N-vector is the gather (the only inter-GPU traffic besides the barrier).
    sync()
    pk = [st.peaks() for st in streams]
    peaks_abs = sharding.gather_stream_values(my_streams, [p_[1] for p_ in pk], S * world, dist, red_dev)

    # The same loop once more, long enough for the GPU's clocks to settle (a 20-step region is over in
    # 13 ms): reported beside `value`, never instead of it.
    steady = None
    if args.steps < 300 and world == 1:
        nlong = 400
        for _ in range(50):
            plan.run()
        sync()
        watch = PowerWatch(dev)
        tl = time.perf_counter()
        with watch:
            for _ in range(nlong):
                plan.run()
            sync()
        dl = (time.perf_counter() - tl) / nlong
        steady = {"steps": nlong, "ms_per_step": round(dl * 1e3, 4), "msamples_per_s": round(S * T * P * C / dl / 1e6, 1),
                  "power": watch.summary(),
                  "note": "same launches, 400 steps after 50 more warm-up steps: the timed region above is too short "
                          "for the clocks to settle"}

    frames_per_step_gpu = S * T * P
    frames_total = frames_per_step_gpu * world * args.steps
    msamples = frames_total * C / dt / 1e6
    mframes = frames_total / dt / 1e6
    units_per_launch = S * C * T                         # block-channels one launch processes
    ab = alg_bytes(P, K, S)
    tb = tiled_bytes(P, K, T)

    # per-kernel durations: HIP events on the engine's own stream, over the same loop
    eng.set_profiling(True)
    eng.reset_profile()
    for _ in range(max(args.steps, 200)):
        plan.run()
    sync()
    prof = eng.get_profile()
    eng.set_profiling(False)
    kms = {k: v["ms"] / max(1, v["launches"]) for k, v in prof.items()}
    dominant = max(kms, key=kms.get)
    # HBM bytes per launch: PMC counters cannot be read from inside this process, so they come from
    # the committed rocprofv3 --pmc passes of this same command (profiles/traffic.json, written by
    # tools/profile.sh with the profile's tag and its kernel-trace averages).  The entry is used only
    # if this run's kernel time agrees with the profiled run's (15 %): other code, other bytes.
    shape_key = "S%d_T%d_K%d_C%d" % (S, T, K, C)
    tj = {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
        except Exception:
            tj = {}
    entry = tj.get(shape_key) or {}
    launched = eng.last_kernels()
    traffic = (entry.get("bytes") or {}).get(dominant)
    traffic_note = None
    if traffic is not None:
        ok, traffic_note = profile_applies(entry, launched, kms, 0.15, 0.0, roles=[dominant])
        if not ok:
            sys.stderr.write("bench.py: WARNING " + str(traffic_note) + "\n")
            traffic = None
    applies = {k: profile_applies(entry, launched, kms, 0.15, 0.0, roles=[k])[0] for k in kms}   # per kernel: same name, same time
    achieved = (traffic / (kms[dominant] * 1e-3) / 1e9) if traffic else None
    frac_alg = ab[dominant] * units_per_launch / (kms[dominant] * 1e-3) / 1e9 / HBM_PEAK_GBS
    roofline = {"bound": "hbm", "kernel": {"forward": "K1 forward", "mac": "K2 mac", "inverse": "K3 inverse"}[dominant],
                "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                "traffic": traffic, "traffic_source": entry.get("profile"), "traffic_note": traffic_note or entry.get("note"),
                "kernel_ms": round(kms[dominant], 4), "kernels_ms": {k: round(v, 4) for k, v in kms.items()},
                "kernels_launched": launched, "kernel_name": launched.get(dominant),
                "profiled_kernel": (entry.get("kernels") or {}).get(dominant),
                "min_bytes_per_launch": int(tb[dominant] * units_per_launch),
                "frac_of_min_bytes": round(tb[dominant] * units_per_launch / (kms[dominant] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                # never "no roofline": when the committed PMC traffic does not apply to this run (`frac` null, the reason in
                # traffic_note) the fraction by the minimum bytes the launch must move still bounds the true one from below
                "frac_lower_bound": round(tb[dominant] * units_per_launch / (kms[dominant] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "frac_lower_bound_why": ("`frac` is null: " + (traffic_note or "profiles/traffic.json has no entry for shape " + shape_key)
                                         + "; this is min_bytes_per_launch / kernel time / peak, a floor of the true fraction")
                if traffic is None else "counter bytes are in use: `frac` is the measured fraction, this its floor",
                "frac_alg": {"applicable": T == 1, "value": round(frac_alg, 4),
                             "why": "SURVEY.md 8(d)'s streaming formula re-reads K spectra per output block; a "
                                    "run-ahead call re-uses them on chip, so this figure is not a roofline fraction"},
                "all_kernels": {k: {"ms": round(kms[k], 4), "kernel": launched.get(k),
                                    "traffic": (entry.get("bytes") or {}).get(k) if applies[k] else None,
                                    "frac": round((entry.get("bytes") or {}).get(k, 0) / (kms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                    if applies[k] and (entry.get("bytes") or {}).get(k) else None,
                                    "frac_of_min_bytes": round(tb[k] * units_per_launch / (kms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                                for k in kms},
                "path": {"min_bytes_per_block_channel": int(tb["total"]),
                         "frac_of_min_bytes": round(tb["total"] * units_per_launch * world * args.steps / dt / 1e9 / (HBM_PEAK_GBS * world), 4)}}

    # What this GPU's HBM gives ANY kernel, reads and writes apart (fe_engine_hbm_rates: plain 16-byte streaming
    # kernels over 2 GiB, HIP events), and each kernel's time against its own bytes at those two rates one after
    # the other.  The nominal 8 TB/s of `peak` is out of reach of a kernel that writes.
    try:
        rates = eng.hbm_rates2(1 << 31, 20)
        rd, wr = (entry.get("read") or {}), (entry.get("write") or {})
        # Stores care about the address pattern (one front of consecutive kilobytes moving through the buffer: 4.0 - 5.1 TB/s;
        # every workgroup its own contiguous region, as the engine's kernels write: 5.6 - 6.1), loads do not: the model takes
        # the better of the two store rates — what a kernel that writes can get from this GPU.
        wrate = max(rates["write"], rates["write_regions"])
        model = {}
        for k in kms:
            if applies[k] and rd.get(k) and wr.get(k):
                t_model = rd[k] / (rates["read"] * 1e9) + wr[k] / (wrate * 1e9)
                # ... and against the rate of a plain kernel that reads AND writes (a copy, every workgroup its own region):
                # mixed traffic pays for the turn-arounds of the DRAM bus, which the two separate rates do not show
                t_copy = (rd[k] + wr[k]) / (max(rates["copy"], rates["copy_regions"]) * 1e9)
                model[k] = {"model_ms": round(t_model * 1e3, 4), "frac": round(t_model / (kms[k] * 1e-3), 4),
                            "at_copy_rate_ms": round(t_copy * 1e3, 4), "frac_at_copy_rate": round(t_copy / (kms[k] * 1e-3), 4)}
        roofline["measured_hbm"] = {"read_GBs": round(rates["read"], 1), "write_GBs": round(rates["write"], 1),
                                    "copy_GBs": round(rates["copy"], 1),
                                    "write_own_regions_GBs": round(rates["write_regions"], 1),
                                    "copy_own_regions_GBs": round(rates["copy_regions"], 1),
                                    "what": "plain streaming kernels on this GPU in this run: 16 bytes per lane over 2 GiB, "
                                            "20 passes, HIP events (copy counts bytes read + written); write / copy: a grid-stride "
                                            "front (rounds 2 - 3 quoted these), *_own_regions: every workgroup its own contiguous region",
                                    "kernel_time_at_these_rates": model or None,
                                    "frac_meaning": "frac: (PMC read bytes / read rate + PMC write bytes / the better write rate) / measured kernel "
                                                    "time — a floor that ignores read / write interference; frac_at_copy_rate: PMC bytes / the "
                                                    "better copy rate / measured kernel time — against a plain kernel with mixed traffic"}
    except Exception as ex:                                  # a measurement aid: never fails the bench line
        roofline["measured_hbm"] = {"error": str(ex)}

    extras = world == 1 and not args.no_extras
    skip = set(x for x in args.skip.split(",") if x)
    # ---- streaming form (one block per stream per call = SoundProcessor::Process granularity): here K2
    # really streams K spectra per block, so algorithmic and moved bytes coincide ----
    streaming = None
    if extras and "streaming" not in skip:
        st1 = [flt.open_stream(1) for _ in range(S)]
        plan1 = BatchPlan(st1, [x.data_ptr() for x in xs], [y.data_ptr() for y in ys], [P] * S,
                          FE_DEVICE_PTRS | FE_ASYNC)
        for _ in range(K + 2):
            plan1.run()
        sync()
        n1 = max(50, args.steps * 2)
        t1 = time.perf_counter()
        for _ in range(n1):
            plan1.run()
        sync()
        d1 = (time.perf_counter() - t1) / n1
        eng.set_profiling(True)
        eng.reset_profile()
        for _ in range(50):
            plan1.run()
        sync()
        p1 = eng.get_profile()
        eng.set_profiling(False)
        k1ms = {k: v["ms"] / max(1, v["launches"]) for k, v in p1.items()}
        mac1_gbs = ab["mac"] * S * C / (k1ms["mac"] * 1e-3) / 1e9
        e1 = tj.get("S%d_T1_K%d_C%d" % (S, K, C)) or {}
        launched1 = eng.last_kernels()
        ok1, note1 = profile_applies(e1, launched1, k1ms, 0.25, 5.0, roles=["mac"])
        streaming = {"bound": "hbm", "kernel": "K2 mac (one block per call)", "blocks_per_call": 1,
                     "achieved": round(mac1_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(mac1_gbs / HBM_PEAK_GBS, 4), "alg_bytes_per_launch": int(ab["mac"] * S * C),
                     "traffic": (e1.get("bytes") or {}).get("mac") if ok1 else None, "traffic_source": e1.get("profile"), "traffic_note": note1,
                     "kernels_launched": launched1,
                     "kernel_ms": round(k1ms["mac"], 4), "kernels_ms": {k: round(v, 4) for k, v in k1ms.items()},
                     "ms_per_step": round(d1 * 1e3, 4), "msamples_per_s": round(S * P * C / d1 / 1e6, 1),
                     "path_frac": round(ab["total"] * S * C / d1 / 1e9 / HBM_PEAK_GBS, 4)}
        for s_ in st1:
            s_.close()

    # ---- end to end: the same batch from page-locked host buffers, PCIe inside the timed region ----
    end_to_end = None
    if extras and "end_to_end" not in skip:
        try:
            hin = [torch.empty(T * P, C).pin_memory() for _ in range(S)]
            hout = [torch.empty(T * P, C).pin_memory() for _ in range(S)]
            for s in range(S):
                hin[s].copy_(xs[s])
            sync()
            hs = [flt.open_stream(T) for _ in range(S)]
            hplan = BatchPlan(hs, [t_.data_ptr() for t_ in hin], [t_.data_ptr() for t_ in hout], [T * P] * S, FE_HOST_PTRS)
            hplan.run()
            ok = bool(np.allclose(hout[0].numpy(), y1[0], atol=2e-6)) if 0 in y1 else None
            nh = 6
            th = time.perf_counter()
            for _ in range(nh):
                hplan.run()
            dh = (time.perf_counter() - th) / nh
            end_to_end = {"msamples_per_s": round(S * T * P * C / dh / 1e6, 1), "ms_per_step": round(dh * 1e3, 3),
                          "buffers": "page-locked host memory, H2D + kernels + D2H pipelined in chunks of whole streams (>= 64 MB each, up to 32)",
                          "pcie_GBs_each_way": round(S * T * P * C * 4 / dh / 1e9, 1), "matches_resident_run": ok}
            for s_ in hs:
                s_.close()
            del hin, hout
        except Exception as e:  # noqa: BLE001
            end_to_end = {"error": repr(e)}

    # ---- the drop-in call: one synchronous stereo block through fe_stream_process ----
    single = None
    if extras and "single_block" not in skip:
        try:
            L = fa.lib()
            nbytes = P * C * 4
            buf = ctypes.c_void_p()
            assert L.fe_host_alloc(nbytes, ctypes.byref(buf)) == 0
            st = flt.open_stream(1)
            assert L.fe_stream_bind_host_buffer(st.h, buf, nbytes) == 0
            arr = np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(ctypes.c_float)), shape=(P * C,))
            arr[:] = np.random.default_rng(9).uniform(-1, 1, P * C).astype(np.float32)

            def loop(n, in_p, out_p, stream):
                t_ = time.perf_counter()
                for _ in range(n):
                    rc = L.fe_stream_process(stream.h, in_p, P, out_p, None, None)
                    assert rc == 0
                return (time.perf_counter() - t_) / n
            loop(K + 20, buf, buf, st)
            zc = min(loop(200, buf, buf, st) for _ in range(3))
            st2 = flt.open_stream(1)
            a_in = np.random.default_rng(9).uniform(-1, 1, P * C).astype(np.float32)
            a_out = np.zeros(P * C, np.float32)
            pi, po = a_in.ctypes.data_as(ctypes.c_void_p), a_out.ctypes.data_as(ctypes.c_void_p)
            loop(K + 20, pi, po, st2)
            staged = min(loop(200, pi, po, st2) for _ in range(3))
            single = {"single_block_us": round(zc * 1e6, 1), "staged_pageable_us": round(staged * 1e6, 1),
                      "what": "fe_stream_process: one synchronous 8192-frame stereo block, K = %d, host pointers; "
                              "first figure with the block buffer page-locked and bound to the stream (what "
                              "folve::SoundProcessor does), second with ordinary memory (staged copies)" % K,
                      "realtime_factor": round(P / FS / zc, 0)}
            st.close(); st2.close()
            L.fe_host_free(buf)
        except Exception as e:  # noqa: BLE001
            single = {"error": repr(e)}

    # ---- the drop-in call under load: N file threads, each its own folve::SoundProcessor pulling single blocks as
    # ConvolveFileHandler does — a C++ host over include/folve_host.h (tools/dropin/dropin_threads.cpp, built by
    # __graft_entry__.build()), run as a child process; the same filter through the real loader (.conf + WAV) ----
    drop_in = None
    if extras and "drop_in" not in skip:
        try:
            import subprocess
            import tempfile
            exe = os.path.join(ROOT, "tools", "dropin", "dropin_threads")
            if not os.path.exists(exe):
                raise RuntimeError("tools/dropin/dropin_threads not built (python -c 'import __graft_entry__ as g; g.build()')")
            d = tempfile.mkdtemp(prefix="folve_dropin_")
            ir = np.stack(taps, axis=1).astype(np.float64)
            ir16 = np.round(ir / np.abs(ir).max() * 0.9 * 32767).astype("<i2")
            with open(os.path.join(d, "ir.wav"), "wb") as f:           # 16-bit PCM WAV, as the demo filters' impulse files
                data = ir16.tobytes()
                f.write(b"RIFF" + (36 + len(data)).to_bytes(4, "little") + b"WAVEfmt " + (16).to_bytes(4, "little") +
                        (1).to_bytes(2, "little") + (C).to_bytes(2, "little") + (FS).to_bytes(4, "little") +
                        (FS * C * 2).to_bytes(4, "little") + (C * 2).to_bytes(2, "little") + (16).to_bytes(2, "little") +
                        b"data" + len(data).to_bytes(4, "little") + data)
            with open(os.path.join(d, "filter-44100.conf"), "w") as f:
                f.write("/convolver/new %d %d 256 %d\n" % (C, C, size))
                for c in range(C):
                    f.write("/impulse/read %d %d 2e-3 0 0 0 %d ir.wav\n" % (c + 1, c + 1, c + 1))
            runs = []
            # (threads, combiner, run-ahead depth in blocks): depth 1 is the reference's one block per Process() call
            for nt, comb, ra in ((1, 1, 1), (1, 1, 64), (16, 1, 64), (64, 1, 1), (128, 1, 1), (64, 1, 32), (64, 1, 64), (64, 1, 128), (64, 0, 1)):
                # long enough that the run-ahead ramp and the ragged end (threads finishing their last chunks) do not weigh
                nblk = (300 if not comb else 2000) if ra == 1 else (20000 if nt == 1 else 8192 if nt <= 16 else max(4096, 48 * ra))
                r = subprocess.run([exe, os.path.join(d, "filter-44100.conf"), str(nt), str(nblk), str(comb), "json",
                                    "run_ahead=%d" % ra],
                                   stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=180)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                runs.append(json.loads(line[-1]) if line else {"threads": nt, "combiner": bool(comb), "run_ahead": ra,
                                                               "error": "rc %d" % r.returncode})
            # The multi-GPU path folve itself would run: ONE process, ProcessorPool -> DeviceRouter spreading the open files
            # over every visible GPU (least-loaded, sticky), each file thread and its page-locked ring placed on its GPU's
            # NUMA node.  Only when more than one GPU is visible to this process.
            ndev = fa.lib().fe_device_count()
            multi = None
            if ndev > 1:
                per_gpu = 64
                nt = min(per_gpu * ndev, 512)
                env = dict(os.environ)
                env.pop("FOLVE_AMD_DEVICES", None)
                r = subprocess.run([exe, os.path.join(d, "filter-44100.conf"), str(nt), "2048", "1", "json", "run_ahead=64", "pin=1"],
                                   stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=300, env=env)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                multi = json.loads(line[-1]) if line else {"error": "rc %d" % r.returncode}
                multi["what"] = ("one process, %d file threads over %d GPUs through folve::DeviceRouter (streams to the least-loaded "
                                 "GPU, one combiner and one engine per GPU, no collective), run-ahead 64, threads and rings "
                                 "NUMA-placed next to their GPU" % (nt, ndev))
            # cfg5's shape on ONE device: eight router slots (eight engines, combiners and copies of the filter) on this GPU,
            # 512 file threads, 64 per slot — everything of the 8-GPU in-process path except seven more devices and buses.
            # What it shows is that the sharder, the per-slot combiners and 512 threads cost nothing beside one slot's 64
            # threads on the same bus; the 8-GPU rate itself needs the hardware (`drop_in_threads_multi_gpu`).
            cfg5_one = None
            if ndev == 1:
                env = dict(os.environ, FOLVE_AMD_DEVICES="0,0,0,0,0,0,0,0")
                r = subprocess.run([exe, os.path.join(d, "filter-44100.conf"), "512", "2048", "1", "json", "run_ahead=64"],
                                   stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=300, env=env)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                cfg5_one = json.loads(line[-1]) if line else {"error": "rc %d" % r.returncode}
                cfg5_one["what"] = ("cfg5's shape on one device: 512 file threads over 8 router slots (FOLVE_AMD_DEVICES=0,0,0,0,0,0,0,0), "
                                    "64 streams per slot, run-ahead 64; one GPU and one bus carry all eight slots.  NOT a stand-in for the "
                                    "8-GPU rate: engines that share a device share its copy engines and its bus (two slots x 64 threads on "
                                    "one device: 266 k blocks/s against 562 k for one slot x 64), and 512 threads share this box's CPU quota; "
                                    "it shows that the sharder places 64 streams on every slot and that all eight engines, combiners and "
                                    "pipelines run at once")
            # every run against the bus: bytes each way per second, and as a fraction of what `end_to_end` moved in this run
            e2e_gbs = (end_to_end or {}).get("pcie_GBs_each_way")
            for r_ in runs:
                if r_.get("blocks_per_s"):
                    gbs = r_["blocks_per_s"] * P * C * 4 / 1e9
                    r_["pcie_GBs_each_way"] = round(gbs, 2)
                    r_["of_end_to_end"] = round(gbs / e2e_gbs, 3) if e2e_gbs else None
            drop_in = {"what": "N host threads, each a folve::SoundProcessor (page-locked ring, per-GPU combiner) pulling 8192-frame "
                               "stereo blocks as ConvolveFileHandler::AddMoreSoundData does: FillBuffer -> WriteProcessed over "
                               "sf_readf_float / sf_writef_float-shaped callbacks that copy every block in and out, K = %d; "
                               "run_ahead = blocks a processor reads ahead of its reader (1 = the reference's one block per "
                               "Process() call); child process, tools/dropin/dropin_threads.cpp" % K,
                       "usable_cpus": usable_cpus()[0], "runs": runs, "multi_gpu": multi, "cfg5_shape_one_device": cfg5_one}
        except Exception as e:  # noqa: BLE001
            drop_in = {"error": repr(e)}

    # ---- the other single-GPU configurations, each with its own roofline ----
    configs = None
    if extras and "configs" not in skip:
        configs = {}
        for name in OTHER_CONFIGS:
            try:
                configs[name] = config_line(name, args.config_blocks, dev=dev, cpu=not args.no_cpu_baseline)
            except Exception as e:  # noqa: BLE001
                configs[name] = {"error": repr(e)}

    mixed_filters = None
    if extras and "mixed" not in skip:
        try:
            mixed_filters = measure_mixed_filters(dev=dev)
        except Exception as e:  # noqa: BLE001
            mixed_filters = {"error": repr(e)}

    if rank == 0:
        out = {
            "metric": "Msamples/s convolved (44.1k/2ch, 256k-tap) + realtime-stream count; HBM % of peak",
            "value": round(msamples, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg3: %d concurrent 44.1 kHz/%d-ch streams per GPU, %d-tap shared random FIR, "
                                   "P=%d K=%d, %d blocks per stream per step, PCM resident in HBM" % (S, C, size, P, K, T),
                       "streams_per_gpu": S, "total_streams": S * world, "channels": C, "taps": size, "block": P,
                       "partitions": K, "blocks_per_step": T, "pcm": "resident in HBM (see end_to_end for the PCIe-inclusive rate)",
                       "sharding": "streams over GPUs (gpu = stream mod N), no data-path collective"},
            "mframes_per_s": round(mframes, 1),
            "realtime_streams": int(mframes * 1e6 / FS),
            "steady_state": steady,
            "parity_rms": parity_abs, "parity_rel": parity_rel,
            "stream_peaks": {"streams": int(len(peaks_abs)), "max_abs": round(float(peaks_abs.max()), 4), "min_abs": round(float(peaks_abs.min()), 4),
                             "what": "max |output| of every stream of the job, gathered over the ranks (K3's per-stream maxima)"},
            "parity": "first two steps of streams %s vs the float64 linear convolution, gate %.0e" % (check, PARITY_TOL),
            "roofline": roofline,
            "roofline_streaming": streaming,
            "end_to_end": end_to_end,
            "single_block": single,
            "drop_in_threads": drop_in,
            "drop_in_threads_multi_gpu": (drop_in or {}).get("multi_gpu") if isinstance(drop_in, dict) else None,
            "configs": configs,
            "mixed_filters": mixed_filters,
            "cpu_baseline": None,
        }
        if world > 1:
            out["shards"] = [sharding.shard_streams(S * world, world, r) for r in range(world)]
        if dist is not None:
            out["process_group"] = {"backend": backend, "world_size": dist.get_world_size(), "forced_at_world_size_1": force_dist}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the CPU leg runs when every timed GPU region is over and (N > 1) the other ranks are gone: host cores to itself
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_leg(args, P, C, size)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
