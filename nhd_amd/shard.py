"""Node-axis sharding across GPUs: one process per GPU, contiguous blocks of the
candidate order, pods replicated, winners picked by a max-reduction of the packed score words.

Data path on the GPU: the all-reduce runs inside libnhdfit (RCCL, `nhdfit_comm_init`) between the fit and
the mapping kernel.  This module holds the host-side arithmetic around it - shard bounds, the order-
preserving uint64 <-> int64 map needed when the reduction goes through torch.distributed (which has no
uint64 MAX; used by the gloo tests and as a host-side reducer), and the merge of per-rank mappings.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

SIGN = np.uint64(1) << np.uint64(63)


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; blocks are multiples of 64 nodes (one ballot word)
    except the last, so shards never split a feasibility word."""
    chunks = (n + 63) // 64
    per = (chunks + world - 1) // world
    lo = min(n, rank * per * 64)
    hi = min(n, (rank + 1) * per * 64)
    return lo, hi


def to_ordered_int64(score: np.ndarray) -> np.ndarray:
    """uint64 -> int64 such that unsigned order == signed order (flip the top bit)."""
    return (score.astype(np.uint64) ^ SIGN).view(np.int64)


def from_ordered_int64(x: np.ndarray) -> np.ndarray:
    return x.view(np.uint64) ^ SIGN


def allreduce_max_scores(score: np.ndarray, dist=None) -> np.ndarray:
    """Max-reduce packed scores over the default torch.distributed group (any backend)."""
    if dist is None:
        import torch.distributed as dist
    import torch
    t = torch.from_numpy(to_ordered_int64(score).copy())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return from_ordered_int64(t.numpy())


def merge_mappings(maps: np.ndarray, dist=None) -> np.ndarray:
    """Each pod's mapping is valid on exactly one rank (the winner's owner) and all-zero elsewhere."""
    if dist is None:
        import torch.distributed as dist
    import torch
    t = torch.from_numpy(maps.view(np.int8).astype(np.int32).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy().astype(np.int8).view(maps.dtype).reshape(maps.shape)
