"""Block-range sharding of an aggregation across ranks.

Every block column is independent (src/bmaggregator.h:1184-1218), so rank r of W
owns the contiguous block range shard_range(nblocks, r, W); inputs never cross
xGMI and the only exchange is the sum of the per-rank popcounts (8 B per group).
"""
from __future__ import annotations


def shard_range(nblocks: int, rank: int, world: int) -> tuple[int, int]:
    """contiguous, exhaustive, balanced to within one block"""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    q, r = divmod(nblocks, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def allreduce_counts(counts, group=None):
    """sum per-rank counts (a torch int64 tensor) over all ranks: RCCL on GPU, gloo on CPU"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return counts
