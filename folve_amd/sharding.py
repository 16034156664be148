"""Stream sharding across the GPUs of one node (one process per GPU).

The path shards by independent units: a stream's state (FDL ring + input tail)
is private and streams only share the read-only filter, so the multi-GPU form of
folve's ProcessorPool is `gpu = stream_index mod G`, sticky for the stream's
lifetime.  There is no data-path collective; torch.distributed
(RCCL on GPUs, gloo in the CPU tests) carries bookkeeping only: the barrier
around the timed region, the max-over-ranks of the elapsed time, the sum of the
processed units and the gather of per-stream peaks.
"""


def shard_streams(n_total, world, rank):
    """Global stream indices owned by `rank`: round-robin, as the pool hands out processors."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_total, world))


def owner_of(stream_index, world):
    return stream_index % world


def aggregate_throughput(units_local, seconds_local, dist=None, device=None):
    """(total units over all ranks, max seconds over ranks, units per second)."""
    if dist is None or not dist.is_initialized():          # (a group of one still reduces: bench.py's FOLVE_BENCH_FORCE_DIST)
        return units_local, seconds_local, units_local / seconds_local
    import torch
    t = torch.tensor([seconds_local], dtype=torch.float64, device=device)
    u = torch.tensor([float(units_local)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()), float(t.item()), float(u.item()) / float(t.item())


def gather_stream_values(local_indices, local_values, n_total, dist=None, device=None):
    """Every rank's per-stream values (e.g. peaks) in global stream order, on every rank."""
    import torch
    out = torch.zeros(n_total, dtype=torch.float64, device=device)
    if len(local_indices):
        out[torch.as_tensor(local_indices, device=device)] = torch.as_tensor(local_values, dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(out, op=dist.ReduceOp.SUM)      # disjoint shards: sum == union
    return out.cpu().numpy()
