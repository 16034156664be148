"""The CPU baseline legs: the CPU restatement of zita-convolver's algorithm (oracle/) timed on this box's host cores, on
bounded samples.  The ONLY module of bench.py that touches oracle/ — as the reported baseline, never as the product —
and, where the box has it, the real libzita-convolver through tests/compile/zita_ref.cpp."""
import os
import sys

from .power import usable_cpus

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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

def cpu_baseline_leg(cpu_seconds, P, C, size):
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
            zb = max(8, int(0.3 * cpu_seconds / max(1e-4, zita_real.bench(zexe, C, size, cores, 8, cores)["seconds"] / 8.0)))
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
    fv, fs, f1, f1s = timed(lambda ns, nb, nt: O.fast_bench_streams(ns, nb, nt, C, size, 3, native=native), 0.6 * cpu_seconds)
    sv, ss, s1, s1s = timed(lambda ns, nb, nt: O.bench_streams(ns, nb, nt, C, C, size, 3, native=native), 0.4 * cpu_seconds)
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

