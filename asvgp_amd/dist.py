"""Multi-GPU data parallelism for the Phi pass: contiguous N-shards per rank, ONE sum-reduction of the
packed [(k+1) M band | M D rhs | 1 yy] fp64 buffer (+ the row count) over RCCL (torch.distributed 'nccl' backend on
ROCm) - or gloo on CPU in the tests.  The reference has no distributed code; the statistics are plain sums over
datapoints (gpr.py:41-44) which is what makes this exact up to fp64 summation order."""
import torch
import torch.distributed as dist


def shard_bounds(N, world_size, rank):
    """contiguous, balanced [lo, hi) shard of N rows; even shard starts keep the 16-B vector-load path aligned."""
    per = (N + world_size - 1) // world_size
    per += per & 1
    lo = min(rank * per, N)
    hi = min(lo + per, N)
    return lo, hi


def allreduce_stats(stats, n_local, group=None):
    """Sum the packed statistics buffer and the local row count across ranks.  Returns global N (python int).
    One collective for the payload (98 312 B at M=2048, k=4); the count rides in a second 8-byte all-reduce."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return int(n_local)
    dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    n = torch.tensor([float(n_local)], dtype=torch.float64, device=stats.device)
    dist.all_reduce(n, op=dist.ReduceOp.SUM, group=group)
    return int(round(n.item()))
