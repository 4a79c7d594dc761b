"""Thin object wrapper over the C-ABI: one :class:`Engine` = one nhdfit_ctx = one GPU."""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np

from . import _lib, pack

SCORE_MASK = 0x7FFFFFFFFFFFFFFF


def _p(a: Optional[np.ndarray]):
    """Address of the array's buffer for a c_void_p parameter (every libnhdfit prototype is declared in _lib._SIGS, so a plain
    int converts; half the cost of .ctypes.data_as - this sits on FindNode's per-pod path).  The caller keeps `a` alive."""
    return None if a is None else a.ctypes.data


class Engine:
    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        h = ctypes.c_void_p()
        rc = self.lib.nhdfit_create(device, ctypes.byref(h))
        if rc != 0:
            raise _lib.NhdFitError(rc, (self.lib.nhdfit_last_error(None) or b"?").decode())
        self.ctx = h
        self.device = device
        self.n = 0
        self.P = 0
        self.global_base = 0
        self._dict_version = -1
        self.n_wide = 0

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.nhdfit_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _chk(self, rc):
        _lib.check(self.ctx, rc)

    # ---- state ------------------------------------------------------------------------
    def set_dictionary(self, packer: pack.Packer):
        if self._dict_version == packer.dict_version:
            return
        caps, sig_off, pool_off, glimit, cc, ncls, nsig, npools, ncc = packer.dictionary_arrays()
        gs = packer.group_set_array()
        self._chk(self.lib.nhdfit_set_dictionary(self.ctx, packer.max_cores_per_numa, packer.max_gpus_per_numa,
                                                 _p(gs), len(packer.group_sets), _p(caps), ncls, _p(sig_off), nsig, _p(pool_off), _p(glimit),
                                                 npools, _p(cc), ncc))
        self._dict_version = packer.dict_version

    def forget_dictionary(self):
        """The next set_dictionary uploads whatever packer it is given (a NEW Packer restarts its version count)."""
        self._dict_version = -1

    def upload(self, table: pack.NodeTable, global_base: int = 0, first: int = 0, capacity: Optional[int] = None):
        """Full upload (first == 0 and nothing uploaded yet) or delta upload of `table` at `first`."""
        cap = max(capacity or 0, first + table.n)
        self._chk(self.lib.nhdfit_reserve_nodes(self.ctx, cap, global_base))
        arrs = [np.ascontiguousarray(x) for x in (table.p0, table.p1, table.p2, table.p3, table.p4, table.detail)]
        self._chk(self.lib.nhdfit_upload_nodes(self.ctx, first, table.n, *[_p(a) for a in arrs]))
        if table.origin is not None and table.n:
            origin = np.ascontiguousarray(table.origin)          # (bound to a name: _p hands out a bare address)
            self._chk(self.lib.nhdfit_upload_origin(self.ctx, first, table.n, _p(origin)))
        sharing = getattr(self, "_share", None) is not None
        whole = first == 0 and table.n >= self.n
        if sharing and not whole and self.n_wide:
            # ENABLE_SHARING, delta upload: device-side commits (wide_commit, schedule_batch) added rx / tx to speed_used records this
            # host copy has never seen - fetch them before nhdfit_wide_upload switches sharing off, so that the nodes OUTSIDE the slice
            # go back up as the device left them (a wide record showing the cores taken beside a speed_used that forgot the claim)
            dev = self.wide_share_download()
            if len(dev):
                self._share[:len(dev)] = dev
        if table.n and (table.wide or self.n_wide):               # nodes beyond the fast layout: their records replace those of this range
            recs = table.wide_records(first)
            self._chk(self.lib.nhdfit_wide_upload(self.ctx, first, table.n, _p(recs) if len(recs) else None, len(recs)))
            self.n_wide = self.wide_count()
        self.n = max(self.n, first + table.n)
        # nhd/Node.py:20 ENABLE_SHARING = True: every node is a wide record and carries its NICs' speed_used (nhdfit_wide_share).  The
        # device wants the records of ALL wide nodes after every wide upload: this engine keeps them by node index
        if table.share:
            if not sharing or len(self._share) < self.n:
                grown = np.zeros(self.n, pack.WIDE_SHARE)
                if sharing:
                    grown[:len(self._share)] = self._share
                self._share = grown
            missing = [i for i in range(table.n) if i not in table.share] if len(table.share) != table.n else []
            if missing:
                raise _lib.NhdFitError(-5, "ENABLE_SHARING: node %d of the uploaded slice carries no speed_used record" % (first + missing[0]))
            for i, rec in table.share.items():
                self._share[first + i] = rec
            if self.n_wide != self.n:
                raise _lib.NhdFitError(-5, "ENABLE_SHARING: a node of the mirror is not held by the general path")
            share = np.ascontiguousarray(self._share[:self.n])
            self._chk(self.lib.nhdfit_wide_share_upload(self.ctx, _p(share), len(share)))
        elif sharing and whole:
            self._share = None                                      # the whole mirror replaced by a table without speed_used records: sharing is off
            self._chk(self.lib.nhdfit_wide_share_upload(self.ctx, None, 0))
        elif sharing and table.n:
            # a slice without speed_used records under ENABLE_SHARING (a node that no longer fits a wide record, a table packed with the
            # constant off): nhdfit_wide_upload has switched the sharing arithmetic off - never continue with the shipped arithmetic silently
            raise _lib.NhdFitError(-5, "ENABLE_SHARING: the uploaded slice [%d, %d) carries no speed_used records" % (first, first + table.n))
        self.global_base = global_base

    def wide_count(self) -> int:
        k = ctypes.c_uint32(0)
        self._chk(self.lib.nhdfit_wide_count(self.ctx, ctypes.byref(k)))
        return int(k.value)

    def wide_download(self) -> np.ndarray:
        """The mirror's wide records (ascending node index) as they are now - after device-side commits."""
        k = ctypes.c_uint32(0)
        out = np.zeros(self.wide_count(), pack.WIDE)
        self._chk(self.lib.nhdfit_wide_download(self.ctx, _p(out) if len(out) else None, len(out), ctypes.byref(k)))
        return out

    def wide_share_download(self) -> np.ndarray:
        """The NICs' speed_used per wide record (ascending node index) as the device's commits left them; empty without ENABLE_SHARING."""
        k = ctypes.c_uint32(0)
        out = np.zeros(self.wide_count(), pack.WIDE_SHARE)
        self._chk(self.lib.nhdfit_wide_share_download(self.ctx, _p(out) if len(out) else None, len(out), ctypes.byref(k)))
        return out[:int(k.value)]

    def wide_commit(self, node: int, req: np.ndarray, mapping: np.ndarray, busy_time: float) -> np.ndarray:
        out = np.zeros((), pack.WIDE_PLACEMENT)
        req = np.ascontiguousarray(req)
        mapping = np.ascontiguousarray(mapping)
        self._chk(self.lib.nhdfit_wide_commit(self.ctx, int(node), _p(req), _p(mapping), float(busy_time), _p(out)))
        return out

    def wide_placements(self) -> np.ndarray:
        """Placements the last schedule_batch made on wide nodes (`pod` = index in that call's batch, `node` = local index)."""
        k = ctypes.c_uint32(0)
        self._chk(self.lib.nhdfit_wide_placements(self.ctx, None, 0, ctypes.byref(k)))
        out = np.zeros(int(k.value), pack.WIDE_PLACEMENT)
        if len(out):
            self._chk(self.lib.nhdfit_wide_placements(self.ctx, _p(out), len(out), ctypes.byref(k)))
        return out

    def apply_deltas(self, deltas: np.ndarray) -> np.ndarray:
        """K3 (nhdfit_apply_deltas): release / reclaim / reset / scalar writes applied to the device mirror in array order.
        `deltas["node"]` = local index.  Returns the per-delta status (pack.DELTA_OK / DELTA_REPACK)."""
        deltas = np.ascontiguousarray(deltas, dtype=pack.DELTA).reshape(-1)
        status = np.zeros(len(deltas), np.uint8)
        if len(deltas):
            self._chk(self.lib.nhdfit_apply_deltas(self.ctx, _p(deltas), len(deltas), _p(status)))
        return status

    def reset_nodes(self):
        self._chk(self.lib.nhdfit_set_node_count(self.ctx, 0))
        self.n = 0
        self.n_wide = 0

    def set_outputs(self, bitmap=True, mapping=True):
        self._chk(self.lib.nhdfit_set_outputs(self.ctx, int(bitmap), int(mapping)))

    # ---- one-shot ---------------------------------------------------------------------
    def find(self, reqs: np.ndarray, now: float, cand: Optional[np.ndarray] = None, want_bitmap=True, want_map=True):
        reqs = np.ascontiguousarray(reqs)
        P = len(reqs)
        score = np.zeros(P, np.uint64)
        bitmap = np.zeros(((self.n + 63) // 64, P), np.uint64) if want_bitmap else None
        maps = np.zeros(P, pack.MAPPING) if want_map else None
        if cand is not None:
            cand = np.ascontiguousarray(cand, dtype=np.uint64)
            assert cand.shape == ((self.n + 63) // 64,)
        self._chk(self.lib.nhdfit_find(self.ctx, _p(reqs), P, float(now), _p(cand), _p(score), _p(bitmap), _p(maps)))
        self.P = P
        return score, bitmap, maps

    def find_sequential(self, reqs: np.ndarray, now: float, cand: Optional[np.ndarray] = None):
        """Mode B (sequential commit inside the batch): (node index or -1, mappings, status) per pod."""
        reqs = np.ascontiguousarray(reqs)
        P = len(reqs)
        node = np.zeros(P, np.int64)
        maps = np.zeros(P, pack.MAPPING)
        status = np.zeros(P, np.int32)
        if cand is not None:
            cand = np.ascontiguousarray(cand, dtype=np.uint64)
        self._chk(self.lib.nhdfit_find_sequential(self.ctx, _p(reqs), P, float(now), _p(cand), _p(node), _p(maps), _p(status)))
        self.P = P
        return node, maps, status

    def schedule_batch(self, reqs: np.ndarray, now: float, packer: pack.Packer, cand: Optional[np.ndarray] = None, apply: bool = True):
        """Mode B with the commit step on the device (nhdfit_schedule_batch): (node index or -1, mappings, placements,
        status) per pod.  apply=True leaves the commits in the device mirror.  A commit that puts a node into a NIC
        state the dictionary has no signature for stops the device pass; the state is interned here (`packer`), the
        node's plane 3 patched, and the pods that are left go out as a new batch (the mirror holds the placements made
        so far) - with Packer.close_signatures() up front this never happens."""
        reqs = np.ascontiguousarray(reqs)
        P = len(reqs)
        node = np.zeros(P, np.int64)
        maps = np.zeros(P, pack.MAPPING)
        places = np.zeros(P, pack.PLACEMENT)
        status = np.zeros(P, np.int32)
        if cand is not None:
            cand = np.ascontiguousarray(cand, dtype=np.uint64)
        first = 0
        self.last_wide_places = {}                                  # pod index of THIS call -> nhdfit_wide_placement (pods that landed on wide nodes)
        while first < P:
            done = ctypes.c_uint32(0)
            self._chk(self.lib.nhdfit_schedule_batch(self.ctx, _p(reqs[first:]), P - first, float(now), _p(cand), int(apply),
                                                     _p(node[first:]), _p(maps[first:]), _p(places[first:]), _p(status[first:]),
                                                     ctypes.byref(done)))
            last = first + done.value
            if self.n_wide:
                for wp in self.wide_placements():
                    self.last_wide_places[first + int(wp["pod"])] = wp.copy()
            # (vectorised: a Python loop over the pods here was 0.25-0.4 ms per 4 096 pods - a tenth of the whole call)
            stuck = (node[first:last][status[first:last] == pack.COMMIT_NEW_SIG] - self.global_base).tolist()
            if last < P and not (apply and stuck):
                raise _lib.NhdFitError(-5, "the device stopped a sequential batch early without a NIC state to intern")
            patched = []
            for v in stuck if apply else []:
                one = self.download(v, 1)
                sn, sp = packer.sigs_from_detail(one.detail[0])
                one.p3[0]["sig_numa"] = sn
                one.p3[0]["sig_pci"] = sp
                patched.append((v, one))
            if patched:
                self.set_dictionary(packer)
                for v, one in patched:
                    self.upload(one, global_base=self.global_base, first=v, capacity=self.n)
            first = last
        self.P = P
        return node, maps, places, status

    def commit(self, node: int, req: np.ndarray, mapping: np.ndarray, busy_time: float) -> np.ndarray:
        """The commit step for one placement (nhdfit_commit): updates the device mirror, returns the placement record."""
        out = np.zeros((), pack.PLACEMENT)
        req = np.ascontiguousarray(req)
        mapping = np.ascontiguousarray(mapping)
        self._chk(self.lib.nhdfit_commit(self.ctx, int(node), _p(req), _p(mapping), float(busy_time), _p(out)))
        return out

    # ---- pods with 5..8 processing groups: the general path over every node (nhdfit_big_find / nhdfit_big_commit) --------
    def big_find(self, reqs: np.ndarray, now: float, cand: Optional[np.ndarray] = None, want_map=True):
        """Mode A for big requests (pack.BIG_REQ): (scores, mappings pack.BIG_MAPPING)."""
        reqs = np.ascontiguousarray(reqs, dtype=pack.BIG_REQ)
        P = len(reqs)
        score = np.zeros(P, np.uint64)
        maps = np.zeros(P, pack.BIG_MAPPING) if want_map else None
        if cand is not None:
            cand = np.ascontiguousarray(cand, dtype=np.uint64)
            assert cand.shape == ((self.n + 63) // 64,)
        if P:
            self._chk(self.lib.nhdfit_big_find(self.ctx, _p(reqs), P, float(now), _p(cand), _p(score), _p(maps)))
        return score, maps

    def big_commit(self, node: int, req: np.ndarray, mapping: np.ndarray, busy_time: float) -> np.ndarray:
        """The commit step of a big request on node `node` (ordinary or wide): updates the mirror, returns pack.BIG_PLACEMENT."""
        out = np.zeros((), pack.BIG_PLACEMENT)
        req = np.ascontiguousarray(req, dtype=pack.BIG_REQ)
        mapping = np.ascontiguousarray(mapping, dtype=pack.BIG_MAPPING)
        self._chk(self.lib.nhdfit_big_commit(self.ctx, int(node), _p(req), _p(mapping), float(busy_time), _p(out)))
        return out

    def download(self, first: int = 0, count: Optional[int] = None) -> pack.NodeTable:
        count = self.n - first if count is None else count
        t = pack.empty_table(count)
        t.origin = None                                     # not read back: uploading this table leaves the origin records alone
        self._chk(self.lib.nhdfit_download_nodes(self.ctx, first, count, _p(t.p0), _p(t.p1), _p(t.p2), _p(t.p3), _p(t.p4), _p(t.detail)))
        if self.n_wide:
            t.wide = {int(w["index"]) - first: w.copy() for w in self.wide_download() if first <= int(w["index"]) < first + count}
        return t

    # ---- pipelined --------------------------------------------------------------------
    def stage(self, reqs: np.ndarray):
        reqs = np.ascontiguousarray(reqs)
        self._chk(self.lib.nhdfit_stage_requests(self.ctx, _p(reqs), len(reqs)))
        self.P = len(reqs)

    def enqueue(self, now: float):
        self._chk(self.lib.nhdfit_enqueue_step(self.ctx, float(now)))

    def sync(self):
        self._chk(self.lib.nhdfit_sync(self.ctx))

    def fetch(self, want_bitmap=False, want_map=True):
        score = np.zeros(self.P, np.uint64)
        bitmap = np.zeros(((self.n + 63) // 64, self.P), np.uint64) if want_bitmap else None
        maps = np.zeros(self.P, pack.MAPPING) if want_map else None
        self._chk(self.lib.nhdfit_fetch(self.ctx, _p(score), _p(bitmap), _p(maps)))
        return score, bitmap, maps

    def stats(self) -> _lib.Stats:
        s = _lib.Stats()
        self._chk(self.lib.nhdfit_get_stats(self.ctx, ctypes.byref(s)))
        return s

    def reset_stats(self):
        self._chk(self.lib.nhdfit_reset_stats(self.ctx))

    # ---- collective -------------------------------------------------------------------
    def unique_id(self) -> bytes:
        buf = ctypes.create_string_buffer(128)
        rc = self.lib.nhdfit_comm_unique_id(buf)
        if rc != 0:
            raise _lib.NhdFitError(rc, (self.lib.nhdfit_last_error(None) or b"?").decode())
        return buf.raw

    def comm_init(self, nranks: int, rank: int, uid: bytes):
        self._chk(self.lib.nhdfit_comm_init(self.ctx, nranks, rank, ctypes.c_char_p(uid)))

    def comm_destroy(self):
        self._chk(self.lib.nhdfit_comm_destroy(self.ctx))

    def comm_rank(self):
        """(rank, ranks) of the context's communicator; (0, 1) without one."""
        r, n = ctypes.c_int(0), ctypes.c_int(1)
        self._chk(self.lib.nhdfit_comm_rank(self.ctx, ctypes.byref(r), ctypes.byref(n)))
        return r.value, n.value

    def comm_sendrecv(self, send: np.ndarray, dst: int, recv: np.ndarray, src: int) -> None:
        """One grouped ncclSend / ncclRecv over the context's communicator (nhdfit_comm_sendrecv): `send` goes to rank `dst`,
        `recv` is filled from rank `src`; either may be None (with its peer -1)."""
        if send is not None:
            send = np.ascontiguousarray(send)
        if recv is not None and not recv.flags["C_CONTIGUOUS"]:
            raise ValueError("recv buffer must be contiguous")
        self._chk(self.lib.nhdfit_comm_sendrecv(self.ctx, send.ctypes.data if send is not None else None, send.nbytes if send is not None else 0, int(dst),
                                                recv.ctypes.data if recv is not None else None, recv.nbytes if recv is not None else 0, int(src)))

    def comm_allreduce_sum_u8(self, buf: np.ndarray) -> None:
        """In-place element-wise sum of a contiguous uint8 array over the ranks (nhdfit_comm_allreduce_sum_u8)."""
        if buf.dtype != np.uint8 or not buf.flags["C_CONTIGUOUS"]:
            raise ValueError("a contiguous uint8 array")
        self._chk(self.lib.nhdfit_comm_allreduce_sum_u8(self.ctx, buf.ctypes.data, buf.nbytes))


class _ShardView(Engine):
    """An Engine whose context belongs to a group (the group destroys it)."""

    def __init__(self, lib, ctx, device):  # noqa: D401 - no nhdfit_create here
        self.lib = lib
        self.ctx = ctypes.c_void_p(ctx)
        self.device = device
        self.n = 0
        self.P = 0
        self.global_base = 0
        self._dict_version = -1
        self.n_wide = 0

    def close(self):
        self.ctx = None


class GroupEngine:
    """One process, several GPUs: the node axis of the mirror cut into contiguous shards (multiples of 64 nodes), one
    context per device, winners picked by ONE all-reduce(max) of the packed scores over xGMI inside libnhdfit
    (nhdfit_group_find).  Presents the Engine interface HipMatcher uses, so `HipMatcher(devices=[0, 1, ...])` lets the
    reference's unmodified one-thread scheduler (nhd/NHDScheduler.py:43,50,277) drive all GPUs of the node.
    `engine_factory` (tests only): build the shards from host-twin engines and reduce on the host instead."""

    def __init__(self, devices, engine_factory=None):
        self.devices = list(devices)
        if not self.devices:
            raise ValueError("GroupEngine needs at least one device")
        self.n = 0
        self.P = 0
        self.global_base = 0
        self.group = None
        self._bounds = []
        self.last_wide_places = {}
        if engine_factory is not None:
            self.lib = None
            self.shards = [engine_factory(d) for d in self.devices]
        else:
            self.lib = _lib.load()
            g = ctypes.c_void_p()
            devs = (ctypes.c_int * len(self.devices))(*self.devices)
            rc = self.lib.nhdfit_group_create(devs, len(self.devices), ctypes.byref(g))
            if rc != 0:
                raise _lib.NhdFitError(rc, (self.lib.nhdfit_last_error(None) or b"?").decode())
            self.group = g
            self.shards = [_ShardView(self.lib, self.lib.nhdfit_group_ctx(g, k), d) for k, d in enumerate(self.devices)]

    def close(self):
        if self.group is not None:
            self.lib.nhdfit_group_destroy(self.group)
            self.group = None
        for s in self.shards:
            s.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # ---- state ------------------------------------------------------------------------
    def set_dictionary(self, packer: pack.Packer):
        for s in self.shards:
            s.set_dictionary(packer)

    def forget_dictionary(self):
        for s in self.shards:
            s.forget_dictionary()

    def reset_nodes(self):
        for s in self.shards:
            s.reset_nodes()
        self.n = 0
        self._bounds = []

    def _shard_of(self, i: int) -> int:
        for k, (lo, hi) in enumerate(self._bounds):
            if lo <= i < hi:
                return k
        raise IndexError(i)

    def upload(self, table: pack.NodeTable, global_base: int = 0, first: int = 0, capacity: Optional[int] = None):
        """Full upload (first == 0: defines the shard bounds) or delta upload of a run of nodes at `first`."""
        from .sharding import shard_bounds
        if first == 0 and (not self._bounds or table.n >= self.n):
            n = table.n
            self._bounds = [shard_bounds(n, len(self.shards), r) for r in range(len(self.shards))]
            for s, (lo, hi) in zip(self.shards, self._bounds):
                s.reset_nodes()
                if hi > lo:
                    s.upload(table.slice(lo, hi), global_base=lo)
                else:
                    s.global_base = lo
            self.n = n
            self._has_gpu = self._gpu_flags(table)
            return
        self._has_gpu[first:first + table.n] = self._gpu_flags(table)
        lo_i = first
        while lo_i < first + table.n:                               # a run may straddle shards
            k = self._shard_of(lo_i)
            lo, hi = self._bounds[k]
            hi_i = min(first + table.n, hi)
            self.shards[k].upload(table.slice(lo_i - first, hi_i - first), global_base=lo, first=lo_i - lo, capacity=hi - lo)
            lo_i = hi_i

    @staticmethod
    def _gpu_flags(table: pack.NodeTable) -> np.ndarray:
        """len(Node.gpus) > 0 per node of `table` (Matcher.SelectNode's preference, nhd/Matcher.py:401-413): the planes' flag - and,
        for a wide node, its record's GPU count (its entry in the planes is a placeholder without flags)."""
        has = np.array((table.p2["flags"] & pack.NF_HAS_GPU) != 0)
        for i, rec in (table.wide or {}).items():
            has[i] = int(rec["n_gpus"]) > 0
        return has

    def set_outputs(self, bitmap=True, mapping=True):
        for s in self.shards:
            s.set_outputs(bitmap, mapping)

    def _nogpu_words(self, k: int) -> np.ndarray:
        """Shard k's nodes without a GPU installed, one bit per node (static per node: the hardware)."""
        lo, hi = self._bounds[k]
        bits = np.zeros(((hi - lo + 63) // 64) * 64, np.uint8)
        bits[:hi - lo] = ~self._has_gpu[lo:hi]
        return np.packbits(bits, bitorder="little").view(np.uint64).copy()

    # ---- one-shot ---------------------------------------------------------------------
    def find(self, reqs: np.ndarray, now: float, cand: Optional[np.ndarray] = None, want_bitmap=False, want_map=True):
        reqs = np.ascontiguousarray(reqs)
        P = len(reqs)
        cands = None
        if cand is not None:
            cand = np.ascontiguousarray(cand, dtype=np.uint64)
            cands = [np.ascontiguousarray(cand[lo // 64:(hi + 63) // 64]) if hi > lo else None for lo, hi in self._bounds]
        self.P = P
        if self.group is None:                                      # host twin shards: reduce here
            score = np.zeros(P, np.uint64)
            maps = np.zeros(P, pack.MAPPING)
            parts = []
            for k, s in enumerate(self.shards):
                if self._bounds[k][1] > self._bounds[k][0]:
                    parts.append((k, s.find(reqs, now, cand=None if cands is None else cands[k], want_bitmap=False, want_map=want_map)))
            for k, (sc, _, mp) in parts:
                score = np.maximum(score, sc)
            for k, (sc, _, mp) in parts:
                lo, hi = self._bounds[k]
                idx = np.where(score == 0, -1, (SCORE_MASK - (score & np.uint64(SCORE_MASK))).astype(np.int64))
                own = (sc == score) & (idx >= lo) & (idx < hi)
                if want_map:
                    maps[own] = mp[own]
            return score, None, (maps if want_map else None)
        score = np.zeros(P, np.uint64)
        maps = np.zeros(P, pack.MAPPING) if want_map else None
        owner = np.zeros(P, np.int32)
        cptr = None
        if cands is not None:                                       # (`cands` keeps the per-shard arrays alive across the call)
            cptr = (ctypes.c_void_p * len(self.shards))(*[None if c is None else c.ctypes.data for c in cands])
        rc = self.lib.nhdfit_group_find(self.group, _p(reqs), P, float(now), cptr, _p(score), _p(maps), _p(owner))
        if rc != 0:
            raise _lib.NhdFitError(rc, (self.lib.nhdfit_group_last_error(self.group) or b"?").decode())
        return score, None, maps

    def schedule_batch(self, reqs: np.ndarray, now: float, packer: pack.Packer, cand: Optional[np.ndarray] = None, apply: bool = True):
        """Mode B over all shards, exactly the one-by-one loop's decisions (nhd/NHDScheduler.py:425-437 with
        Matcher.SelectNode's order, nhd/Matcher.py:401-413): a pod takes the first feasible node of the cluster, a pod
        without GPUs the first feasible GPU-less node if there is one.  Two facts make that separable by shard:
          * nodes without GPUs only ever receive pods without GPUs, so their state evolves under that subsequence alone:
            the GPU-less pods walk the shards' GPU-less nodes in shard order, each shard's sequential pass
            (nhdfit_schedule_batch, candidates = its GPU-less nodes) handing the pods it could not place to the next;
          * whatever is left of them and the pods with GPUs then walk the shards in the same manner over all nodes:
            a pod reaches shard s iff no node of shards < s could take it at its turn, and shard s's state depends
            only on the pods placed there before.  (A GPU-less node that refused a pod only lost resources since.)
        Each pass runs on ONE device; a pod costs a pass only on the shards it is offered to.  apply=False restores the
        shards from copies taken up front."""
        reqs = np.ascontiguousarray(reqs)
        P = len(reqs)
        node = np.full(P, -1, np.int64)
        maps = np.zeros(P, pack.MAPPING)
        places = np.zeros(P, pack.PLACEMENT)
        status = np.zeros(P, np.int32)
        if cand is not None:
            cand = np.ascontiguousarray(cand, dtype=np.uint64)
        live = [k for k, (lo, hi) in enumerate(self._bounds) if hi > lo]
        saved = {} if apply else {k: self.shards[k].download(0, self._bounds[k][1] - self._bounds[k][0]) for k in live}
        nogpu = {k: self._nogpu_words(k) for k in live}
        wants_gpu = reqs["gpus"].sum(axis=1) > 0
        touched = {}
        self.last_wide_places = {}

        def offer(pods: np.ndarray, gpu_less_nodes_only: bool) -> np.ndarray:
            for k in live:
                if len(pods) == 0:
                    break
                lo, hi = self._bounds[k]
                mask = None if cand is None else np.ascontiguousarray(cand[lo // 64:(hi + 63) // 64])
                if gpu_less_nodes_only:
                    mask = nogpu[k] if mask is None else mask & nogpu[k]
                    if not mask.any():
                        continue
                nd, mp, pl, st = self.shards[k].schedule_batch(reqs[pods], now, packer, cand=mask, apply=True)
                for j, wp in getattr(self.shards[k], "last_wide_places", {}).items():   # pods that landed on a wide node of this shard
                    wp = wp.copy()
                    wp["node"] = int(wp["node"]) + lo                                    # (global node index, as `node` below)
                    self.last_wide_places[int(pods[j])] = wp
                got = nd >= 0
                node[pods[got]] = nd[got]
                maps[pods[got]] = mp[got]
                places[pods[got]] = pl[got]
                status[pods[got]] = st[got]
                if got.any():
                    a, b = int(nd[got].min()) - lo, int(nd[got].max()) - lo + 1
                    touched[k] = (min(a, touched[k][0]), max(b, touched[k][1])) if k in touched else (a, b)
                pods = pods[~got]
            return pods

        left = offer(np.flatnonzero(~wants_gpu), True)
        rest = np.sort(np.concatenate([np.flatnonzero(wants_gpu), left]))
        offer(rest, False)
        if not apply:
            for k, (a, b) in touched.items():
                lo, hi = self._bounds[k]
                self.shards[k].upload(saved[k].slice(a, b), global_base=lo, first=a, capacity=hi - lo)
        self.P = P
        return node, maps, places, status

    def find_sequential(self, reqs: np.ndarray, now: float, cand: Optional[np.ndarray] = None, packer: Optional[pack.Packer] = None):
        if packer is None:
            raise ValueError("GroupEngine.find_sequential needs the packer (NIC states are interned on the way)")
        node, maps, _, status = self.schedule_batch(reqs, now, packer, cand=cand, apply=False)
        return node, maps, status

    def commit(self, node: int, req, mapping, busy_time):
        k = self._shard_of(node)
        return self.shards[k].commit(node - self._bounds[k][0], req, mapping, busy_time)

    def wide_commit(self, node: int, req, mapping, busy_time):
        k = self._shard_of(node)
        return self.shards[k].wide_commit(node - self._bounds[k][0], req, mapping, busy_time)

    def big_find(self, reqs: np.ndarray, now: float, cand: Optional[np.ndarray] = None, want_map=True):
        """Big requests over every shard: each device runs the general pass on its nodes (nhdfit_big_find), the score words -
        which carry the global node index - are max-merged and the owner's mapping kept.  (A rare path: the merge of
        len(shards) x P words is done here, not by a collective.)"""
        reqs = np.ascontiguousarray(reqs, dtype=pack.BIG_REQ)
        P = len(reqs)
        score = np.zeros(P, np.uint64)
        maps = np.zeros(P, pack.BIG_MAPPING)
        parts = []
        for k, s in enumerate(self.shards):
            lo, hi = self._bounds[k]
            if hi > lo:
                mask = None if cand is None else np.ascontiguousarray(np.ascontiguousarray(cand, dtype=np.uint64)[lo // 64:(hi + 63) // 64])
                parts.append((k, s.big_find(reqs, now, cand=mask, want_map=want_map)))
        for k, (sc, mp) in parts:
            score = np.maximum(score, sc)
        idx = np.where(score == 0, -1, (SCORE_MASK - (score & np.uint64(SCORE_MASK))).astype(np.int64))
        for k, (sc, mp) in parts:
            lo, hi = self._bounds[k]
            own = (sc == score) & (idx >= lo) & (idx < hi)
            if want_map:
                maps[own] = mp[own]
        return score, (maps if want_map else None)

    def big_commit(self, node: int, req, mapping, busy_time):
        k = self._shard_of(node)
        out = self.shards[k].big_commit(node - self._bounds[k][0], req, mapping, busy_time)
        out["node"] = node
        return out

    @property
    def n_wide(self) -> int:
        return sum(getattr(s, "n_wide", 0) for s in self.shards)

    def apply_deltas(self, deltas: np.ndarray) -> np.ndarray:
        """Deltas with GLOBAL node indices, routed to the shards that own the nodes (order kept per shard)."""
        deltas = np.ascontiguousarray(deltas, dtype=pack.DELTA).reshape(-1)
        status = np.zeros(len(deltas), np.uint8)
        owner = np.array([self._shard_of(int(v)) for v in deltas["node"]], dtype=np.int64)
        for k in np.unique(owner):
            sel = np.flatnonzero(owner == k)
            part = deltas[sel].copy()
            part["node"] -= self._bounds[int(k)][0]
            status[sel] = self.shards[int(k)].apply_deltas(part)
        return status

    def download(self, first: int = 0, count: Optional[int] = None) -> pack.NodeTable:
        count = self.n - first if count is None else count
        out = pack.empty_table(count)
        out.origin = None
        i = first
        while i < first + count:
            k = self._shard_of(i)
            lo, hi = self._bounds[k]
            j = min(first + count, hi)
            part = self.shards[k].download(i - lo, j - i)
            for f in ("p0", "p1", "p2", "p3", "p4", "detail"):
                getattr(out, f)[i - first:j - first] = getattr(part, f)
            for q, w in (part.wide or {}).items():
                out.wide[q + i - first] = w
            i = j
        return out

    def stats(self):
        return self.shards[0].stats()


def winner_index(score: int) -> int:
    return SCORE_MASK - (int(score) & SCORE_MASK)
