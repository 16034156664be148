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

Prints ONE JSON line on rank 0 — under 4 KB (benchlib/line.py; the driver keeps an 8 KB tail): the contract's keys,
  parity_rms   — gate, BEFORE anything is timed: the first two steps' output of two streams against the float64 linear
                 convolution; exit 1 above 1e-5
  roofline     — the dominant kernel against the roof that binds it.  `kernel_ms`: THIS run's, a start / stop event bound
                 to each dispatch (the packet's own begin-to-end, what rocprofv3's kernel trace prints).  `frac` = HBM bytes
                 the kernel really moves (rocprofv3 PMC passes of the SAME command, profiles/traffic.json, taken only for
                 the same kernels by name at times that agree) / kernel_ms / 8 TB/s; `k2_valu`: K2's issued flops against
                 the FP32 vector peak; `bound` = the larger of the two
  cpu_baseline — the CPU restatement of zita-convolver's algorithm (oracle/, rebuilt -march=native on this box) on all
                 usable cores and on one, on a bounded sample (rank 0; at N > 1 after the process group is gone)
  configs      — cfg1 / cfg2 / cfg4 / a 2 x 2 filter matrix at 256-block calls: one rate, one fraction, one parity each
and `details`: bench_details.json beside this file, which holds everything else — per-kernel tables of every
configuration, steady_state (400 more steps with socket power and shader clock), roofline_streaming (one block per call:
the streaming formula of SURVEY.md 8(d) applies), end_to_end (PCIe inside the timed region), single_block, drop_in_threads
(1 .. 128 file threads, each its own folve::SoundProcessor), mixed_filters, the CPU legs per configuration.

The pieces live in benchlib/ (formulas, power, profiles, configs, headline, dropin, cpu, line)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.formulas import FS, HBM_PEAK_GBS, PARITY_TOL, alg_bytes, conv_f64, rms, tiled_bytes   # noqa: E402,F401
from benchlib.power import PowerWatch, usable_cpus                                                   # noqa: E402,F401
from benchlib.profiles import norm_kernel, profile_applies, traffic_key                               # noqa: E402,F401
from benchlib.configs import OTHER_CONFIGS                                                            # noqa: E402,F401
from benchlib import line as bench_line                                                               # noqa: E402


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


def parse_args():
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
    ap.add_argument("--skip", default="", help="comma-separated extra legs to leave out: streaming,end_to_end,single_block,drop_in,configs,mixed,hbm")
    ap.add_argument("--only-config", default="", choices=["", "cfg1", "cfg2", "cfg4", "matrix"],
                    help="run only this configuration's loop and print its `configs` entry (what tools/profile_all.sh profiles)")
    ap.add_argument("--config-blocks", type=int, default=256, help="blocks per call of the cfg2 / cfg4 legs")
    ap.add_argument("--details", default="", help="where to write the details file (default: bench_details.json beside bench.py)")
    return ap.parse_args()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    from benchlib import configs as bench_configs
    from benchlib.headline import parse_tune
    if args.only_config:
        import torch  # noqa: F401
        # (--skip longer: without the 1 024-block leg, whose launches can share a kernel and a grid with the 256-block ones —
        # a profile of this command must not average the two)
        print(json.dumps({args.only_config: bench_configs.config_line(args.only_config, args.config_blocks, steps=min(args.steps, 300),
                                                                      tune=parse_tune(args.tune),
                                                                      longer_calls="longer" not in args.skip.split(","))}))
        return

    import torch
    from folve_amd import sharding
    from benchlib import cpu as bench_cpu, dropin as bench_dropin, headline as bench_headline

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
    # FOLVE_BENCH_FORCE_DIST=1: a process group even at world size 1, so that the RCCL branch (init, barrier, the
    # reductions of sharding.aggregate_throughput beside the engine's own HIP streams) runs on a one-GPU box
    force_dist = world == 1 and os.environ.get("FOLVE_BENCH_FORCE_DIST", "") == "1"
    if world > 1 or force_dist:
        dist = bench_headline.init_process_group(world, rank, dev, backend, force_dist)
    assert args.gpus == world, "--gpus must equal WORLD_SIZE (launch N>1 with torch.distributed.run)"
    red_dev = torch.device("cuda", dev) if backend == "nccl" else torch.device("cpu")

    h = bench_headline.Headline(args, world, rank, dev, dist, red_dev)
    S, T, C, size, P, K = h.S, h.T, h.C, h.size, h.P, h.K
    parity_abs, parity_rel, check = h.gate_or_exit()
    dt = h.timed()
    peaks_abs = h.stream_peaks()
    steady = h.steady_state() if (args.steps < 300 and world == 1) else None

    frames_total = S * T * P * world * args.steps
    msamples = frames_total * C / dt / 1e6
    mframes = frames_total / dt / 1e6
    roofline = h.roofline(dt)

    extras = world == 1 and not args.no_extras
    skip = set(x for x in args.skip.split(",") if x)
    if extras and "hbm" not in skip:
        roofline["measured_hbm"] = h.measured_hbm()
    streaming = h.streaming() if (extras and "streaming" not in skip) else None
    end_to_end = bench_dropin.end_to_end(h) if (extras and "end_to_end" not in skip) else None
    single = bench_dropin.single_block(h) if (extras and "single_block" not in skip) else None
    drop_in = None
    if extras and "drop_in" not in skip:
        drop_in = bench_dropin.drop_in_threads(h, (end_to_end or {}).get("pcie_GBs_each_way"))

    # ---- the other single-GPU configurations, each with its own roofline ----
    configs = None
    if extras and "configs" not in skip:
        configs = {}
        for name in OTHER_CONFIGS:
            try:
                configs[name] = bench_configs.config_line(name, args.config_blocks, dev=dev, cpu=not args.no_cpu_baseline)
            except Exception as e:  # noqa: BLE001
                configs[name] = {"error": repr(e)}
    mixed_filters = None
    if extras and "mixed" not in skip:
        try:
            mixed_filters = bench_configs.measure_mixed_filters(dev=dev)
        except Exception as e:  # noqa: BLE001
            mixed_filters = {"error": repr(e)}

    out = None
    if rank == 0:
        out = {
            "metric": "Msamples/s convolved (44.1k/2ch, 256k-tap) + realtime-stream count; HBM % of peak",
            "value": round(msamples, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg3: %d concurrent 44.1 kHz/%d-ch streams per GPU, %d-tap shared random FIR, "
                                   "P=%d K=%d, %d blocks per stream per step, PCM resident in HBM" % (S, C, size, P, K, T),
                       "streams_per_gpu": S, "total_streams": S * world, "channels": C, "taps": size, "block": P,
                       "partitions": K, "blocks_per_step": T, "pcm": "resident in HBM (end_to_end: the PCIe-inclusive rate)",
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
            out["cpu_baseline"] = bench_cpu.cpu_baseline_leg(args.cpu_seconds, P, C, size)
        bench_line.emit(out, args.details or None)


if __name__ == "__main__":
    main()
