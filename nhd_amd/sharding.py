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


def schedule_batch_sharded(engine, reqs: np.ndarray, now: float, packer, nogpu_words: np.ndarray, dist, apply: bool = True,
                           chunk: int = 512, device: str = "cpu"):
    """Mode B with one process per GPU: every rank holds an :class:`nhd_amd.engine.Engine` with its
    contiguous node shard (global_base set) and calls this collectively with the same `reqs`.  Returns the decisions of the
    one-by-one scheduler loop over the WHOLE cluster (nhd/NHDScheduler.py:425-437 with Matcher.SelectNode's order,
    nhd/Matcher.py:401-413) on every rank: (global node index or -1, mappings, placements, status) per pod.

    The argument is GroupEngine.schedule_batch's, with the shards in different processes.  Nodes without GPUs only ever
    receive pods without GPUs, so those pods walk the shards' GPU-less nodes in shard order (walk 1): a shard's sequential
    pass (nhdfit_schedule_batch with its GPU-less nodes as candidates) hands the pods it could not place to the next rank.
    What is left of them at the last rank, and the pods with GPUs, then walk the shards over all nodes in the same manner
    (walk 2): a pod reaches shard s iff no node of the shards before could take it at its turn, shard s's state depends only
    on the pods placed there before, and a GPU-less node that refused a pod has only lost resources since.

    Both walks are PIPELINED over slices of `chunk` pods in the caller's order: rank k works on slice c while rank k+1 works
    on what rank k left of slice c-1, so the batch costs about one rank's passes over it plus the pipeline's fill, not the sum
    of all ranks' passes (a shard's decisions for a slice depend only on the earlier slices' pods offered to it - the order of
    the walk inside a shard is kept).  Walk 1 of every slice runs first on each rank, then walk 2: the two touch disjoint node
    sets except for the left-over GPU-less pods, which walk 2 receives from the last rank before it starts on their slice.

    Traffic: the pods a rank could not place travel to the next rank as ONE fixed-size int32 tensor per slice and walk
    (`dist.isend` / `dist.recv`, 4 * (chunk + 1) bytes); the results meet in ONE `dist.all_reduce(SUM)` of a byte buffer at
    the end - every pod is placed by at most one rank, all others contribute zeros.  No pickled objects.  `dist` is
    torch.distributed (gloo: `device="cpu"`; nccl = RCCL: `device="cuda"`).  `nogpu_words`: this shard's nodes without a GPU
    installed, one bit per node ([chunks] uint64 words).  apply=False restores this rank's shard afterwards."""
    import torch
    from . import pack as _pack
    rank, world = dist.get_rank(), dist.get_world_size()
    reqs = np.ascontiguousarray(reqs)
    P = len(reqs)
    chunk = max(1, int(chunk))
    node1 = np.zeros(P, np.int64)                        # global node index + 1; 0 = not placed by this rank
    maps = np.zeros(P, _pack.MAPPING)
    places = np.zeros(P, _pack.PLACEMENT)
    status = np.zeros(P, np.int32)
    wants_gpu = reqs["gpus"].sum(axis=1) > 0
    saved = None if apply or engine.n == 0 else engine.download(0, engine.n)
    touched = None
    mask_nogpu = np.ascontiguousarray(nogpu_words, dtype=np.uint64)
    slices = [np.arange(a, min(P, a + chunk), dtype=np.int64) for a in range(0, P, chunk)]
    pending = []                                          # (request, tensor) of sends in flight: both stay alive until waited for

    def run(pods: np.ndarray, gpu_less_nodes_only: bool) -> np.ndarray:
        """This shard's sequential pass over `pods` (ascending); returns the pods it could not place."""
        nonlocal touched
        mask = mask_nogpu if gpu_less_nodes_only else None
        if len(pods) == 0 or engine.n == 0 or (mask is not None and not mask.any()):
            return pods
        nd, mp_, pl, st = engine.schedule_batch(reqs[pods], now, packer, cand=mask, apply=True)
        got = nd >= 0
        idx = pods[got]
        node1[idx], maps[idx], places[idx], status[idx] = nd[got] + 1, mp_[got], pl[got], st[got]
        if got.any():
            a, b = int(nd[got].min()) - engine.global_base, int(nd[got].max()) - engine.global_base + 1
            touched = (a, b) if touched is None else (min(a, touched[0]), max(b, touched[1]))
        return pods[~got]

    def send(dst: int, pods: np.ndarray) -> None:
        buf = torch.full((chunk + 1,), -1, dtype=torch.int32)
        buf[0] = len(pods)
        if len(pods):
            buf[1:1 + len(pods)] = torch.from_numpy(pods.astype(np.int32))
        buf = buf.to(device)
        pending.append((dist.isend(buf, dst), buf))

    def recv(src: int) -> np.ndarray:
        buf = torch.empty(chunk + 1, dtype=torch.int32, device=device)
        dist.recv(buf, src)
        host = buf.cpu().numpy()
        return host[1:1 + int(host[0])].astype(np.int64)

    # walk 1: pods without GPUs over the nodes without GPUs, slice by slice down the ranks
    left_over = []                                        # rank 0 only: what the last rank could not place, per slice
    for sl in slices:
        pods = sl[~wants_gpu[sl]] if rank == 0 else recv(rank - 1)
        left = run(pods, True)
        if rank + 1 < world:
            send(rank + 1, left)
        elif world > 1:
            send(0, left)                                 # the last rank's left-overs start walk 2 at rank 0
        else:
            left_over.append(left)
    # walk 2: pods with GPUs and the left-overs of walk 1, in the caller's order, over all nodes
    for c, sl in enumerate(slices):
        if rank == 0:
            left = left_over[c] if world == 1 else recv(world - 1)
            pods = np.sort(np.concatenate([sl[wants_gpu[sl]], left]))
        else:
            pods = recv(rank - 1)
        left = run(pods, False)
        if rank + 1 < world:
            send(rank + 1, left)
    for req, _ in pending:
        req.wait()

    # every pod was placed by at most one rank: the element-wise sum of the ranks' (zero-initialised) results is the result
    if world > 1:
        parts = [node1.view(np.uint8), maps.view(np.uint8).reshape(-1), places.view(np.uint8).reshape(-1), status.view(np.uint8)]
        flat = torch.from_numpy(np.concatenate(parts)).to(device)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        host = flat.cpu().numpy()
        at = 0
        for arr in (node1, maps, places, status):
            nbytes = arr.nbytes
            arr.view(np.uint8).reshape(-1)[:] = host[at:at + nbytes]
            at += nbytes
    if saved is not None and touched is not None:
        a, b = touched
        engine.upload(saved.slice(a, b), global_base=engine.global_base, first=a, capacity=engine.n)
    return node1 - 1, maps, places, status
