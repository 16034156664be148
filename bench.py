#!/usr/bin/env python3
"""bench.py — throughput of the folve convolution hot path on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
64 concurrent synthetic 44.1 kHz / 2-channel streams per GPU through one shared
2-path 262 144-tap random FIR (P = 8192, K = 32), PCM resident in HBM.  One
"step" = one batched pass of the hot path (K1 forward FFT -> K2 MAC -> K3
inverse FFT) over `--blocks` consecutive 8192-frame blocks of every stream
(run-ahead batches, the BufferThread role in folve).  N > 1 shards independent
streams across GPUs (64 per GPU, cfg5 = 512 streams on 8): no data-path
collective, torch.distributed (RCCL) only for the barrier and the max-over-ranks.

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline     — HBM roofline of the dominant kernel (K2 MAC), HIP-event timed
  cpu_baseline — the CPU restatement of zita-convolver's algorithm (oracle/),
                 timed on this box's host cores on a bounded sample (N = 1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FS = 44100


def alg_bytes(P, K_eff, S):
    """SURVEY.md §8(d) algorithmic bytes per block-channel and its per-kernel split."""
    fwd = 8 * P + 8 * (P + 1)                 # read x(n-1), x(n); write one spectrum
    mac = 8 * (P + 1) * K_eff + 8 * (P + 1) * K_eff / S   # read K spectra + the shared filter
    inv = 4 * P                               # write P samples
    return {"forward": fwd, "mac": mac, "inverse": inv, "total": fwd + mac + inv}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--streams", type=int, default=64, help="streams per GPU")
    ap.add_argument("--blocks", type=int, default=64, help="consecutive blocks per stream per step (run-ahead depth)")
    ap.add_argument("--taps", type=int, default=262144)
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU baseline sample length")
    args = ap.parse_args()

    import torch
    import folve_amd as fa
    from folve_amd.capi import BatchPlan, FE_ASYNC, FE_DEVICE_PTRS

    from folve_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # FOLVE_BENCH_DEVICE / FOLVE_BENCH_BACKEND exist so that the N > 1 code path can be exercised on a
    # one-GPU box (all ranks on one device, gloo); the driver's runs use one rank per GPU over RCCL.
    dev = int(os.environ.get("FOLVE_BENCH_DEVICE", local_rank if world > 1 else 0))
    backend = os.environ.get("FOLVE_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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
    flt = fa.Filter(eng, C, C, size)
    P, K = flt.block_size, flt.partitions
    rng = np.random.default_rng(3)
    for c in range(C):                               # one shared filter, C diagonal paths, unit L2 norm
        h = rng.standard_normal(size).astype(np.float32)
        h /= np.linalg.norm(h)
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

    for _ in range(args.warmup):
        plan.run()
    sync(); barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.run()
    sync(); barrier(); sync()
    dt = time.perf_counter() - t0
    _, dt, _ = sharding.aggregate_throughput(S * T * P * args.steps, dt, dist, red_dev)   # max over ranks

    frames_per_step_gpu = S * T * P
    frames_total = frames_per_step_gpu * world * args.steps
    msamples = frames_total * C / dt / 1e6
    mframes = frames_total / dt / 1e6
    units_per_launch = S * C * T                         # block-channels one launch processes
    ab = alg_bytes(P, K, S)

    # per-kernel durations: HIP events on the engine's own stream, over the same loop
    eng.set_profiling(True)
    eng.reset_profile()
    for _ in range(args.steps):
        plan.run()
    sync()
    prof = eng.get_profile()
    eng.set_profiling(False)
    kms = {k: v["ms"] / max(1, v["launches"]) for k, v in prof.items()}
    dominant = max(kms, key=kms.get)
    dom_bytes = ab[dominant] * units_per_launch
    achieved = dom_bytes / (kms[dominant] * 1e-3) / 1e9
    # HBM bytes per launch of the dominant kernel: PMC counters cannot be read from inside this
    # process, so the value comes from the committed rocprofv3 --pmc passes of this same command
    # (profiles/traffic.json, made by tools/profile.sh); null for shapes that were not profiled.
    traffic, tj = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get("S%d_T%d_K%d_C%d" % (S, T, K, C), {}).get(dominant)
        except Exception:
            traffic, tj = None, None
    roofline = {"bound": "hbm", "kernel": "K2 " + dominant if dominant == "mac" else dominant,
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "alg_bytes_per_launch": int(dom_bytes), "kernel_ms": round(kms[dominant], 4),
                "kernels_ms": {k: round(v, 4) for k, v in kms.items()},
                "path": {"alg_bytes_per_block_channel": int(ab["total"]),
                         "achieved": round(ab["total"] * units_per_launch * world * args.steps / dt / 1e9, 1),
                         "frac": round(ab["total"] * units_per_launch * world * args.steps / dt / 1e9 / (HBM_PEAK_GBS * world), 4)}}

    # streaming form (one block per stream per call = SoundProcessor::Process granularity): here K2
    # really streams K spectra per block, so algorithmic and measured bytes coincide
    streaming = None
    if world == 1:
        st1 = [flt.open_stream(1) for _ in range(S)]
        plan1 = BatchPlan(st1, [x.data_ptr() for x in xs], [y.data_ptr() for y in ys], [P] * S,
                          FE_DEVICE_PTRS | FE_ASYNC)
        for _ in range(K + 2):
            plan1.run()
        sync()
        n1 = max(50, args.steps * 4)
        t1 = time.perf_counter()
        for _ in range(n1):
            plan1.run()
        sync()
        d1 = (time.perf_counter() - t1) / n1
        eng.set_profiling(True)
        eng.reset_profile()
        for _ in range(20):
            plan1.run()
        sync()
        p1 = eng.get_profile()
        eng.set_profiling(False)
        mac1_ms = p1["mac"]["ms"] / max(1, p1["mac"]["launches"])
        mac1_gbs = ab["mac"] * S * C / (mac1_ms * 1e-3) / 1e9
        streaming = {"blocks_per_call": 1, "ms_per_step": round(d1 * 1e3, 4),
                     "msamples_per_s": round(S * P * C / d1 / 1e6, 1),
                     "path_frac": round(ab["total"] * S * C / d1 / 1e9 / HBM_PEAK_GBS, 4),
                     "mac_kernel_ms": round(mac1_ms, 4), "mac_achieved_GBs": round(mac1_gbs, 1),
                     "mac_frac": round(mac1_gbs / HBM_PEAK_GBS, 4),
                     "mac_traffic": (tj or {}).get("S%d_T1_K%d_C%d" % (S, K, C), {}).get("mac")}
        for s_ in st1:
            s_.close()

    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        from oracle import oracle as O      # CPU restatement: the baseline being reported, not the product
        cores = os.cpu_count() or 1
        nthreads = cores
        nstreams = nthreads
        # a short probe runs ~2.5x faster per block than the steady state (cold DRAM working set of
        # 8 MB per stream builds up), so size the sample from a 16-block probe
        tprobe = O.bench_streams(nstreams, 16, nthreads, C, C, size, 3) / 16.0    # seconds per block round
        nblocks = int(max(8, min(4096, args.cpu_seconds / max(tprobe * 1.5, 1e-4))))
        tcpu = O.bench_streams(nstreams, nblocks, nthreads, C, C, size, 3)
        cpu = {"value": round(nstreams * nblocks * P * C / tcpu / 1e6, 2), "unit": "Msamples/s", "cores": nthreads,
               "kind": "port",
               "what": "CPU restatement of zita-convolver's algorithm (zita-convolver/FFTW unavailable offline)",
               "sample": "%d streams x %d blocks x %d ch, %d taps, one Convproc per stream, %d threads, %.1f s"
                         % (nstreams, nblocks, C, size, nthreads, tcpu)}

    if rank == 0:
        out = {
            "metric": "Msamples/s convolved (44.1k/2ch, 256k-tap) + realtime-stream count; HBM % of peak",
            "value": round(msamples, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg3: %d concurrent 44.1 kHz/%d-ch streams per GPU, %d-tap shared random FIR, "
                                   "P=%d K=%d, %d blocks per stream per step, PCM resident in HBM" % (S, C, size, P, K, T),
                       "streams_per_gpu": S, "channels": C, "taps": size, "block": P, "partitions": K,
                       "blocks_per_step": T, "sharding": "streams over GPUs, no data-path collective"},
            "mframes_per_s": round(mframes, 1),
            "realtime_streams": int(mframes * 1e6 / FS),
            "roofline": roofline,
            "cpu_baseline": cpu,
            "streaming": streaming,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
