"""Node-axis sharding across GPUs: contiguous blocks of the candidate order, pods replicated,
winners picked by a max-reduction of the packed score words (inside libnhdfit: RCCL, `nhdfit_comm_init` for one
process per GPU, `nhdfit_group_find` for one process driving several GPUs).  This module holds the host-side
arithmetic around it: shard bounds, the order-preserving uint64 <-> int64 map (for reducers without uint64 MAX), and mode B
with one process per GPU (`schedule_batch_sharded`) over a plain transport object - `RcclTransport` is the product's: the
communicator of the rank's own context behind the C-ABI (nhdfit_comm_sendrecv / nhdfit_comm_allreduce_sum_u8: RCCL over xGMI).
Nothing here imports torch.
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


class RcclTransport:
    """Rank-to-rank traffic over the communicator of this rank's context (`Engine.comm_init`): ncclSend / ncclRecv and
    ncclAllReduce behind the C-ABI.  The interface `schedule_batch_sharded` needs of any transport:
        rank, world                              ints
        sendrecv(send, dst, recv, src)           `send` (contiguous ndarray) to rank dst while `recv` is filled from rank src;
                                                 every rank calls it at the same point of the algorithm (a ring step)
        allreduce_sum_u8(buf)                    in-place element-wise sum of a uint8 array over the ranks
    (tests drive the same algorithm over gloo with workload.dist.TorchTransport and the host twin per shard)."""

    def __init__(self, engine):
        self.engine = engine
        self.rank, self.world = engine.comm_rank()

    def sendrecv(self, send: np.ndarray, dst: int, recv: np.ndarray, src: int) -> None:
        self.engine.comm_sendrecv(send, dst, recv, src)

    def allreduce_sum_u8(self, buf: np.ndarray) -> None:
        self.engine.comm_allreduce_sum_u8(buf)


def schedule_batch_sharded(engine, reqs: np.ndarray, now: float, packer, nogpu_words: np.ndarray, transport=None, apply: bool = True,
                           chunk: int = 512):
    """Mode B with one process per GPU: every rank holds an :class:`nhd_amd.engine.Engine` with its
    contiguous node shard (global_base set) and calls this collectively with the same `reqs`.  Returns the decisions of the
    one-by-one scheduler loop over the WHOLE cluster (nhd/NHDScheduler.py:425-437 with Matcher.SelectNode's order,
    nhd/Matcher.py:401-413) on every rank: (global node index or -1, mappings, placements, status) per pod.

    The argument is GroupEngine.schedule_batch's, with the shards in different processes.  Nodes without GPUs only ever
    receive pods without GPUs, so those pods walk the shards' GPU-less nodes in shard order (walk 1): a shard's sequential
    pass (nhdfit_schedule_batch with its GPU-less nodes as candidates) hands the pods it could not place to the next rank.
    What is left of them at the last rank, and the pods with GPUs, then walk the shards over all nodes in the same manner
    (walk 2): a pod reaches shard s iff no node of the shards before could take it at its turn, shard s's state depends only
    on the pods placed there before, and a GPU-less node that refused a pod has only lost resources since - walk 2 therefore
    only ever places on nodes with GPUs, walk 1 on nodes without: the two walks never meet on a node.

    A RING IN LOCK STEP over slices of `chunk` pods in the caller's order.  The 2 * world stations of a slice are rank 0 ..
    world - 1 in walk 1, then rank 0 .. world - 1 in walk 2; slice c is at station j in tick c + j.  In every tick a rank
    runs its walk-1 station (slice t - rank) and its walk-2 station (slice t - rank - world), then ONE exchange: what both
    passes left goes to the next rank while the previous rank's arrives (`transport.sendrecv`, one fixed-size int32 buffer of
    two slots - the last rank's walk-1 left-overs are rank 0's walk-2 input, merged in the caller's order with the slice's
    pods with GPUs).  Rank k thus works on slice c while rank k + 1 works on what rank k left of slice c - 1: the batch costs
    about one rank's passes over it plus the ring's fill, not the sum of all ranks' passes; and every rank makes the same
    sequence of exchanges whatever the pods do - nothing to dead-lock on, nothing to match up by hand (round 4 sent tensors
    with torch.distributed.isend from inside this package).  The results meet in ONE `transport.allreduce_sum_u8` of a byte
    buffer - every pod is placed by at most one rank, all others contribute zeros.  `transport`: see RcclTransport (default:
    the engine's own communicator).  `nogpu_words`: this shard's nodes without a GPU installed, one bit per node ([chunks]
    uint64 words).  apply=False restores this rank's shard afterwards.  Pods placed on nodes beyond the fast layout (wide nodes: 3-4
    sockets, more than 64 cores per socket) come back with `places[pod]["status"] == COMMIT_WIDE`; their placement records are in
    `engine.last_wide_places[pod]` on every rank afterwards."""
    from . import pack as _pack
    if transport is None:
        transport = RcclTransport(engine)
    rank, world = int(transport.rank), int(transport.world)
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
    # does ANY shard hold nodes beyond the fast layout?  Agreed on collectively before the first tick (one byte through the results'
    # all-reduce): the ranks must size the final exchange alike
    flag = np.array([1 if getattr(engine, "n_wide", 0) else 0], np.uint8)
    if world > 1:
        transport.allreduce_sum_u8(flag)
    carry_wide = bool(flag[0])
    wide = np.zeros(P if carry_wide else 0, _pack.WIDE_PLACEMENT)      # per pod: the wide placement record of the rank that placed it (zeros: none)
    mask_nogpu = np.ascontiguousarray(nogpu_words, dtype=np.uint64)
    slices = [np.arange(a, min(P, a + chunk), dtype=np.int64) for a in range(0, P, chunk)]
    S = len(slices)
    empty = np.zeros(0, np.int64)

    def run(pods: np.ndarray, gpu_less_nodes_only: bool) -> np.ndarray:
        """This shard's sequential pass over `pods` (ascending); returns the pods it could not place."""
        nonlocal touched
        mask = mask_nogpu if gpu_less_nodes_only else None
        if len(pods) == 0 or engine.n == 0 or (mask is not None and not mask.any()):
            return pods
        nd, mp_, pl, st = engine.schedule_batch(reqs[pods], now, packer, cand=mask, apply=True)
        if carry_wide and (pl["status"] == _pack.COMMIT_WIDE).any():           # (a placement's own status field says so, as in HipMatcher._run_checked)
            # pods that landed on nodes beyond the fast layout: their physical ids are the owning rank's wide placement records
            # (Engine.last_wide_places, keyed by the position in THIS pass's batch) - kept by the caller's pod index and carried to
            # every rank with the results' all-reduce (round 5 raised here, in the middle of the lock-step ring)
            for k, wp in getattr(engine, "last_wide_places", {}).items():
                wide[int(pods[int(k)])] = wp
        got = nd >= 0
        idx = pods[got]
        node1[idx], maps[idx], places[idx], status[idx] = nd[got] + 1, mp_[got], pl[got], st[got]
        if got.any():
            a, b = int(nd[got].min()) - engine.global_base, int(nd[got].max()) - engine.global_base + 1
            touched = (a, b) if touched is None else (min(a, touched[0]), max(b, touched[1]))
        return pods[~got]

    slot = chunk + 1                                      # a slot: count, then the pods (caller's indices)
    out_buf = np.zeros(2 * slot, np.int32)
    in_buf = np.zeros(2 * slot, np.int32)

    def put(k: int, pods: np.ndarray) -> None:
        out_buf[k * slot] = len(pods)
        out_buf[k * slot + 1:k * slot + 1 + len(pods)] = pods

    def get(k: int) -> np.ndarray:
        return in_buf[k * slot + 1:k * slot + 1 + int(in_buf[k * slot])].astype(np.int64)

    in1 = in2 = empty                                     # what the previous rank left of the slices this rank meets next
    for t in range(S + 2 * world - 1):
        left1 = left2 = empty
        c1, c2 = t - rank, t - rank - world
        if 0 <= c1 < S:                                   # walk 1: pods without GPUs over the nodes without GPUs
            sl = slices[c1]
            left1 = run(sl[~wants_gpu[sl]] if rank == 0 else in1, True)
        if 0 <= c2 < S:                                   # walk 2: pods with GPUs and walk 1's left-overs, in the caller's order, over all nodes
            sl = slices[c2]
            left2 = run(np.sort(np.concatenate([sl[wants_gpu[sl]], in1])) if rank == 0 else in2, False)
        put(0, left1)
        put(1, left2)
        transport.sendrecv(out_buf, (rank + 1) % world, in_buf, (rank - 1) % world)
        in1, in2 = get(0), get(1)                         # (rank 0: slot 0 = the last rank's walk-1 left-overs; its slot 1 - pods no node takes - is dropped)

    # every pod was placed by at most one rank: the element-wise sum of the ranks' (zero-initialised) results is the result
    if world > 1:
        parts = [node1.view(np.uint8), maps.view(np.uint8).reshape(-1), places.view(np.uint8).reshape(-1), status.view(np.uint8)]
        if carry_wide:
            parts.append(wide.view(np.uint8).reshape(-1))
        flat = np.ascontiguousarray(np.concatenate(parts))
        transport.allreduce_sum_u8(flat)
        at = 0
        for arr in (node1, maps, places, status) + ((wide,) if carry_wide else ()):
            nbytes = arr.nbytes
            arr.view(np.uint8).reshape(-1)[:] = flat[at:at + nbytes]
            at += nbytes
    if saved is not None and touched is not None:
        a, b = touched
        engine.upload(saved.slice(a, b), global_base=engine.global_base, first=a, capacity=engine.n)
    # the placements on wide nodes, by the caller's pod index, on EVERY rank - as Engine.schedule_batch leaves them on one
    # (pack.expand_wide_placement turns one into the physical ids; its `node` field is the owner's LOCAL index)
    engine.last_wide_places = {int(i): wide[i].copy() for i in np.flatnonzero(places["status"] == _pack.COMMIT_WIDE)} if carry_wide else {}
    return node1 - 1, maps, places, status
