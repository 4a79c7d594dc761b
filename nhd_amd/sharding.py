"""Node-axis sharding across GPUs: contiguous blocks of the candidate order, pods replicated,
winners picked by a max-reduction of the packed score words (inside libnhdfit: RCCL, `nhdfit_comm_init` for one
process per GPU, `nhdfit_group_find` for one process driving several GPUs).  This module holds the host-side
arithmetic around it: shard bounds and the order-preserving uint64 <-> int64 map (for reducers without uint64 MAX).
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
