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


def schedule_batch_sharded(engine, reqs: np.ndarray, now: float, packer, nogpu_words: np.ndarray, dist, apply: bool = True):
    """Mode B with one process per GPU: every rank holds an :class:`nhd_amd.engine.Engine` with its
    contiguous node shard (global_base set) and calls this collectively with the same `reqs`.  Returns the decisions of the
    one-by-one scheduler loop over the WHOLE cluster (nhd/NHDScheduler.py:425-437 with Matcher.SelectNode's order,
    nhd/Matcher.py:401-413) on every rank: (global node index or -1, mappings, placements, status) per pod.

    Exactly GroupEngine.schedule_batch's argument, with the shards in different processes: nodes without GPUs only ever
    receive pods without GPUs, so those pods first walk the shards' GPU-less nodes in shard order - each shard's sequential
    pass (nhdfit_schedule_batch with its GPU-less nodes as candidates) hands the pods it could not place to the next rank;
    whatever is left of them and the pods with GPUs then walk the shards over all nodes in the same manner (a pod reaches
    shard s iff no node of the shards before could take it at its turn; shard s's state depends only on the pods placed
    there before).  The hand-over is a broadcast of the remaining pod list and of the owner's results from the rank that just
    ran - `dist` is torch.distributed (gloo or nccl) or anything with get_rank / get_world_size / broadcast_object_list.
    `nogpu_words`: this shard's nodes without a GPU installed, one bit per node ([chunks] uint64 words).
    apply=False restores this rank's shard afterwards."""
    from . import pack as _pack
    rank, world = dist.get_rank(), dist.get_world_size()
    reqs = np.ascontiguousarray(reqs)
    P = len(reqs)
    node = np.full(P, -1, np.int64)
    maps = np.zeros(P, _pack.MAPPING)
    places = np.zeros(P, _pack.PLACEMENT)
    status = np.zeros(P, np.int32)
    wants_gpu = reqs["gpus"].sum(axis=1) > 0
    saved = None if apply or engine.n == 0 else engine.download(0, engine.n)
    touched = None

    def offer(pods: np.ndarray, gpu_less_nodes_only: bool) -> np.ndarray:
        nonlocal touched
        for k in range(world):
            box = [None]
            if rank == k:
                got_idx = np.zeros(0, np.int64)
                res = None
                mask = np.ascontiguousarray(nogpu_words, dtype=np.uint64) if gpu_less_nodes_only else None
                if len(pods) and engine.n and (mask is None or mask.any()):
                    nd, mp_, pl, st = engine.schedule_batch(reqs[pods], now, packer, cand=mask, apply=True)
                    got = nd >= 0
                    got_idx = pods[got]
                    res = (nd[got], mp_[got], pl[got], st[got])
                    if got.any():
                        a, b = int(nd[got].min()) - engine.global_base, int(nd[got].max()) - engine.global_base + 1
                        touched = (a, b) if touched is None else (min(a, touched[0]), max(b, touched[1]))
                box = [(got_idx, res)]
            dist.broadcast_object_list(box, src=k)
            got_idx, res = box[0]
            if res is not None and len(got_idx):
                node[got_idx], maps[got_idx], places[got_idx], status[got_idx] = res
                pods = pods[~np.isin(pods, got_idx)]
        return pods

    left = offer(np.flatnonzero(~wants_gpu), True)
    offer(np.sort(np.concatenate([np.flatnonzero(wants_gpu), left])), False)
    if saved is not None and touched is not None:
        a, b = touched
        engine.upload(saved.slice(a, b), global_base=engine.global_base, first=a, capacity=engine.n)
    return node, maps, places, status
