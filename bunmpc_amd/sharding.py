"""Batch sharding across the GPUs of one node: the problems are independent,
so rank r of R owns the contiguous slab [r*B, (r+1)*B) and the data path has NO collective.
torch.distributed (RCCL on GPUs, gloo in the CPU tests) only reduces timing / telemetry
scalars and, when a consumer wants them in one place, gathers the output slabs."""
import numpy as np


def shard_range(rank, world, per_rank):
    """absolute problem indices owned by `rank` (weak scaling: per_rank fixed)"""
    return rank * per_rank, (rank + 1) * per_rank


def reduce_telemetry(dist, torch, device, elapsed, kern_ms, counts):
    """MAX over ranks of the timings, SUM over ranks of the counters; returns python floats."""
    t = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=device)
    c = torch.tensor(list(counts), dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return [float(v) for v in t], [float(v) for v in c]


def gather_slabs(dist, torch, device, slab):
    """all_gather of equally sized per-rank output slabs (numpy in, numpy out, rank order)."""
    x = torch.as_tensor(np.ascontiguousarray(slab), device=device)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return x.cpu().numpy()
    out = [torch.empty_like(x) for _ in range(dist.get_world_size())]
    dist.all_gather(out, x)
    return torch.cat(out, dim=0).cpu().numpy()
