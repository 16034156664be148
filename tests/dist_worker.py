"""Worker for tests/test_sharding_gloo.py: one rank of a world_size-2 gloo job on CPU.

Runs the multi-GPU bookkeeping of bench.py (shard by stream, barrier, max-over-ranks
time, gather of per-stream peaks) with the CPU oracle standing in for the GPU engine —
legal here because this is a test of the sharding logic, not of the product path."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from folve_amd import sharding  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    out_path, n_total = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = sharding.shard_streams(n_total, world, rank)
    size = 300
    h = (np.random.default_rng(3).standard_normal(size) / 10).astype(np.float32)
    peaks, sums = [], []
    dist.barrier()
    t0 = time.perf_counter()
    for s in mine:
        c = O.Convproc(1, 1, size)
        c.impdata_create(0, 0, h, 0)
        sp = O.SoundProcessor.wrap(c)
        x = np.random.default_rng(100 + s).uniform(-1, 1, (1000, 1)).astype(np.float32)
        y = sp.run(x)
        peaks.append(sp.max_output_value())
        sums.append(float(y.astype(np.float64).sum()))
    dist.barrier()
    dt = time.perf_counter() - t0 + 0.01 * rank           # make the ranks' clocks differ
    units, tmax, rate = sharding.aggregate_throughput(len(mine) * 1000, dt, dist)
    allpeaks = sharding.gather_stream_values(mine, peaks, n_total, dist)
    allsums = sharding.gather_stream_values(mine, sums, n_total, dist)
    if rank == 0:
        json.dump({"units": units, "tmax": tmax, "rate": rate, "dt0": dt, "peaks": allpeaks.tolist(),
                   "sums": allsums.tolist(), "mine": mine}, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
