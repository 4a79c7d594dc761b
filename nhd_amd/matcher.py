"""HipMatcher - drop-in for ``nhd.Matcher.Matcher`` (nhd/Matcher.py:21-63).

    self.matcher = Matcher()                                   nhd/NHDScheduler.py:50
    match = self.matcher.FindNode(filt_nodes, top)             nhd/NHDScheduler.py:277
    match[0] -> node name or None,  match[1] -> {'gpu': (..), 'cpu': (..), 'nic': [(numa, idx), ..]}

Same contract: ``FindNode(nl, top)`` returns ``(name, mapping)`` or the 1-tuple ``(None,)``, never
mutates ``nl``/``top``, breaks ties by ``nl`` iteration order, is called from one thread.  The per-node
Python loops are replaced by one pass of the gfx950 kernels over a packed mirror of the cluster
(DESIGN.md).  If libnhdfit.so or the GPU is missing, construction raises - there is no CPU path.

Two ways to keep the device mirror current:

* stateless (default): every call packs ``nl`` again.  Always right, O(N) Python per call.
* ``attach(nodes)``: ``nodes`` is the scheduler's long-lived ``self.nodes`` dict.  Nodes are packed
  once; afterwards only nodes touched through the reference's own mutators (nhd/Node.py:144, 530,
  587, 644, 663, 843, 308) or whose ``active`` / ``maintenance`` / ``groups`` / ``busy_time`` were
  assigned are re-packed (dirty tracking by swapping in a thin subclass).  ``FindNode(nl, top)`` then
  accepts any subset ``nl`` of ``nodes`` in the same relative order (what InitialNodeFilter
  produces, nhd/NHDScheduler.py:235-247) and turns it into a candidate bitmask.

``FindNodes(nl, tops, pod_groups)`` is the batch form (every pod against one snapshot).
"""
from __future__ import annotations

import logging
import time
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import pack
from ._lib import NhdFitError
from .engine import Engine, GroupEngine, winner_index

_MUTATORS = ("SetPhysicalIdsFromMapping", "RemoveResourcesFromTopology", "AddResourcesFromTopology",
             "ResetResources", "ClaimPodNICResources", "SetBusy", "SetGroups", "SetHugepages", "ParseLabels")
_WATCHED = frozenset(("active", "maintenance", "groups", "busy_time"))
_tracked_cache: Dict[type, type] = {}


def _tracked_class(base: type) -> type:
    """Subclass of the node's own class that reports state changes to the matcher owning it."""
    cls = _tracked_cache.get(base)
    if cls is not None:
        return cls

    def __setattr__(self, key, value):
        base.__setattr__(self, key, value)
        if key in _WATCHED:
            hook = self.__dict__.get("_nhdfit_dirty")
            if hook is not None:
                hook(self, key)

    ns = {"__setattr__": __setattr__}
    for name in _MUTATORS:
        orig = getattr(base, name, None)
        if orig is None:
            continue

        def make(orig, name):
            def wrapper(self, *a, **kw):
                ok = False
                out = None
                try:
                    out = orig(self, *a, **kw)
                    ok = True
                    return out
                finally:
                    hook = self.__dict__.get("_nhdfit_dirty")
                    if hook is not None:
                        # the commit step of AttemptScheduling (nhd/NHDScheduler.py:289-304) is mirrored on the device
                        # instead of re-packing the node: see HipMatcher._on_commit / _on_claim
                        # and so are the release / reclaim / reset paths and SetHugepages (row f2: nhdfit_apply_deltas)
                        done = False
                        owner = self.__dict__["_nhdfit_owner"]
                        if ok and name == "SetPhysicalIdsFromMapping":
                            done = owner._on_commit(self, *a, **kw)
                        elif ok and name == "ClaimPodNICResources":
                            done = owner._on_claim(self, *a, **kw)
                        elif ok and name in ("RemoveResourcesFromTopology", "AddResourcesFromTopology") and out is not False:
                            done = owner._on_topology(self, name, *a, **kw)    # (False: the reference gave up half-way, Node.py:543-545 - re-pack)
                        elif ok and name == "ResetResources":
                            done = owner._on_queued_scalar(self, "reset")
                        elif ok and name == "SetHugepages":
                            done = owner._on_queued_scalar(self, "hugepages")
                        elif ok and name == "SetGroups":
                            done = True                              # the assignment to .groups inside it was seen by __setattr__
                        if not done:
                            hook(self, name)
            wrapper.__name__ = orig.__name__
            return wrapper
        ns[name] = make(orig, name)
    cls = type("Tracked" + base.__name__, (base,), ns)
    _tracked_cache[base] = cls
    return cls


# What list(set(itertools.product(range(2), repeat=k))) gives under the CPython set / tuple-hash behaviour the device-side
# model (nhd_amd/csrc/winner_map.h) reproduces (CPython 3.8 - 3.12, 64-bit).  FindNode's mapping is "bit-identical to the
# reference" relative to the interpreter that would run the reference - i.e. this one: if it orders sets differently
# (another Python implementation, a future change of tuple hashing) the model no longer speaks for it and HipMatcher
# refuses to start instead of returning mappings the reference would not.
_SET_ORDER_PROBES = {
    1: [(0,), (1,)],
    2: [(0, 1), (1, 0), (1, 1), (0, 0)],
    3: [(1, 0, 1), (1, 1, 0), (0, 1, 0), (0, 0, 0), (1, 0, 0), (0, 0, 1), (1, 1, 1), (0, 1, 1)],
    4: [(0, 0, 0, 1), (0, 0, 1, 0), (0, 1, 0, 1), (0, 1, 1, 1), (1, 0, 1, 1), (1, 1, 0, 0), (0, 1, 0, 0), (1, 1, 1, 0),
        (0, 1, 1, 0), (0, 0, 0, 0), (1, 0, 1, 0), (1, 0, 0, 1), (1, 1, 0, 1), (1, 0, 0, 0), (0, 0, 1, 1), (1, 1, 1, 1)],
}
_SET_INTERSECTION_PROBE = [(1, 0, 1), (1, 1, 1), (0, 0, 0), (0, 1, 1)]


def check_interpreter_set_model() -> None:
    import itertools
    for k, want in _SET_ORDER_PROBES.items():
        if list(set(itertools.product(range(2), repeat=k))) != want:
            raise RuntimeError("this Python interpreter iterates sets of int tuples in a different order than the set model of "
                               "libnhdfit (CPython 3.8-3.12): FindNode's NUMA mapping would differ from the reference Matcher's")
    a = set(itertools.product(range(2), repeat=3))
    b = {t for t in a if sum(t) != 1}
    c = {t for t in a if t[0] == 0 or t[2] == 1}
    if list(a & b & c) != _SET_INTERSECTION_PROBE:
        raise RuntimeError("this Python interpreter intersects sets in a different order than the set model of libnhdfit")


class HipMatcher:
    def __init__(self, device: int = 0, clock=time.monotonic, engine_factory=None, devices: Optional[Sequence[int]] = None,
                 strict: bool = False):
        """`devices=[0, 1, ...]`: shard the mirror over several GPUs of this process (engine.GroupEngine: node axis
        cut into contiguous shards, one RCCL all-reduce(max) of the packed scores per call picks the winners) - the
        reference's one-thread scheduler keeps calling FindNode exactly as before.
        `engine_factory(device)` exists for the test-suite only (it injects the host build of the
        kernels' arithmetic so the host logic of this class can be exercised without a GPU); the
        default is the HIP engine, which raises when the library or a gfx950 GPU is missing.
        `strict=False` (default) keeps FindNode's contract - `(name, mapping)` or `(None,)`, never an exception
        (nhd/Matcher.py:47-63) - where the device layout cannot hold something: a node beyond the
        layout's capacities (include/nhdfit.h) never matches and is listed in `self.unmirrored` (name -> reason), a
        request beyond them or a device error makes the call answer `(None,)` (the scheduler leaves the pod pending and
        tries again, nhd/NHDScheduler.py:278-287); each is logged.  `strict=True` raises instead."""
        self.logger = logging.getLogger(__name__)
        self.strict = strict
        self.last_placements: List[Optional[dict]] = []    # physical ids per pod of the last ScheduleBatch (see there)
        self._warned: set = set()
        check_interpreter_set_model()
        if devices is not None:
            self.engine = GroupEngine(devices, engine_factory)
        else:
            self.engine = (engine_factory or Engine)(device)
        self.packer = pack.Packer(strict=strict)
        self.clock = clock
        self._attached: Optional[Dict[str, object]] = None
        self._index: Dict[str, int] = {}
        self._names: List[str] = []
        self._table: Optional[pack.NodeTable] = None
        self._dirty: Dict[str, object] = {}
        self._mirror_foreign = False                       # the mirror holds a dict other than the attached one
        self._reasons: Dict[str, set] = {}
        self._claims: Dict[str, frozenset] = {}
        self._batch_ids: Dict[str, list] = {}              # ScheduleBatch(apply=True): ids the device already committed, per node, in order
        self._deltas: List[np.ndarray] = []                # release / reclaim / reset / SetHugepages waiting for the device, in call order
        self.delta_stats = {"applied": 0, "repacked": 0}
        self._uploaded_ids: Optional[Tuple[int, ...]] = None
        # candidate masks of the filtered dicts seen lately, by (length, first, middle, last name) -> [(names, mask)]: pods of a few
        # node groups take turns in the pending list, each group with its own filtered dict (nhd/NHDScheduler.py:235-247)
        self._last_subset: Optional[Dict[tuple, List[Tuple[List[str], np.ndarray]]]] = None
        self._subset_count = self._subset_names = 0

    # ---- mirror maintenance -------------------------------------------------------------
    def attach(self, nodes: Dict[str, object]) -> None:
        """Mirror `nodes` (the scheduler's self.nodes) persistently and track changes to it."""
        self._attached = nodes
        self._mirror_foreign = False
        # a fresh dictionary: NIC signatures, capacity classes and group sets of nodes that left the cluster (or of states
        # nothing is in any more) do not pile up over the life of a scheduler that re-attaches after node churn
        self.packer = pack.Packer(strict=self.strict)
        self.engine.forget_dictionary()
        for node in nodes.values():
            if type(node) not in _tracked_cache.values():
                node.__class__ = _tracked_class(type(node))
            node.__dict__["_nhdfit_dirty"] = self._mark
            node.__dict__["_nhdfit_owner"] = self
        self._full_upload(nodes)
        self.packer.close_signatures()                 # every NIC state a commit can produce gets its signature now
        self.engine.set_dictionary(self.packer)
        self._dirty.clear()
        self._reasons.clear()
        self._deltas = []
        self._claims = {}

    def detach(self) -> None:
        if self._attached:
            for node in self._attached.values():
                node.__dict__.pop("_nhdfit_dirty", None)
                node.__dict__.pop("_nhdfit_owner", None)
        self._attached = None

    # ---- commit step mirrored on the device (row f1) ---------------------------------------
    def _off_planes(self, name: str) -> bool:
        """The node is not (only) held by the five planes: a wide node (its record is re-uploaded whenever it changes - the
        general path has no delta form) or one no layout holds (never matches; nothing to keep current)."""
        i = self._index.get(name)
        return name in self.packer.unmirrored or (i is not None and self._table is not None and bool(self._table.wide) and i in self._table.wide)

    def _mapping_record(self, mapping, big: bool = False) -> np.ndarray:
        G = len(mapping["gpu"])
        m = np.zeros((), pack.BIG_MAPPING if big or G > pack.MAX_GROUPS else pack.MAPPING)
        m["gpu"][:G] = mapping["gpu"]
        m["cpu"][:G + 1] = mapping["cpu"]
        m["nic_numa"][:G] = [x[0] for x in mapping["nic"]]
        m["nic_idx"][:G] = [x[1] for x in mapping["nic"]]
        m["valid"] = 1
        return m

    def CommitPlacement(self, name: str, top, mapping, busy_time: Optional[float] = None) -> dict:
        """The commit step of AttemptScheduling (SetBusy / SetPhysicalIdsFromMapping / ClaimPodNICResources,
        nhd/NHDScheduler.py:289-304) applied to the DEVICE mirror of attached node `name` (nhdfit_commit); returns
        the physical ids the reference would write into `top` (pack.expand_placement).  The Node object itself is
        not touched."""
        i = self._index[name]
        node = self._attached[name] if self._attached is not None else None
        bt = float(node.busy_time if busy_time is None and node is not None else busy_time)
        if pack.needs_general_path(top):                                                # a pod with 5..8 processing groups (or beyond the hugepage table): the general path's commit step
            return self._commit_big(i, node, self.packer.digest_big(top), self._mapping_record(mapping, big=True), bt)
        cached = getattr(self, "_digest_cache", None)                 # the record FindNode made of this very topology a moment ago (the scheduler
        req = cached[1] if cached is not None and cached[0] is top else self.packer.digest(top)   # commits what it has just matched, nhd/NHDScheduler.py:277-304)
        if self.packer.sharing and int(req["flags"]) & pack.RF_NIC_SPLIT and not self.packer.share_exact:
            # (the cached record was admitted when every speed in the mirror was still a multiple of 2^-20; one that is not arrived since)
            raise pack.UnsupportedNode("a processing group with several RX / TX cores (ENABLE_SHARING), and the mirror's speeds are no longer all "
                                       "multiples of 2^-20 Gb/s: the commit's sum would not be the reference's core-by-core accumulation")
        if self._table is not None and self._table.wide and i in self._table.wide:      # a wide node: the general path's commit step
            place = self.engine.wide_commit(i, req, self._mapping_record(mapping), bt)
            G = int(req["n_groups"])
            cpp = int(node.cores_per_proc) if node is not None else int(self._table.wide[i]["cores_per_proc"])
            U = int(node.sockets) if node is not None else int(self._table.wide[i]["numa_nodes"])
            return pack.expand_wide_placement(place, G, cpp, cpp * U, [int(req["gpus"][g]) for g in range(G)])
        place = self.engine.commit(i, req, self._mapping_record(mapping), bt)
        if int(place["status"]) == pack.COMMIT_NEW_SIG:        # cannot happen after close_signatures(); handled anyway
            one = self.engine.download(i, 1)
            sn, sp = self.packer.sigs_from_detail(one.detail[0])
            one.p3[0]["sig_numa"], one.p3[0]["sig_pci"] = sn, sp
            self.engine.set_dictionary(self.packer)
            self.engine.upload(one, first=i, capacity=len(self._names))
        cpp = int(node.cores_per_proc) if node is not None else 0
        G = int(req["n_groups"])
        return pack.expand_placement(place, G, cpp, cpp * int(node.sockets) if node is not None else 0,
                                     [int(req["gpus"][g]) for g in range(G)])

    def _commit_big(self, i: int, node, req, mapping_rec, busy_time: float) -> dict:
        """nhdfit_big_commit on node i (ordinary or wide) -> the physical ids (pack.expand_wide_placement: a big placement
        carries two mask words per batch whichever form the node is mirrored in)."""
        place = self.engine.big_commit(i, req, mapping_rec, busy_time)
        wide = self._table is not None and self._table.wide and i in self._table.wide
        if not wide and int(place["status"]) == pack.COMMIT_NEW_SIG:    # cannot happen after close_signatures(); handled anyway
            one = self.engine.download(i, 1)
            sn, sp = self.packer.sigs_from_detail(one.detail[0])
            one.p3[0]["sig_numa"], one.p3[0]["sig_pci"] = sn, sp
            self.engine.set_dictionary(self.packer)
            self.engine.upload(one, first=i, capacity=len(self._names))
        G = int(req["n_groups"])
        if node is not None:
            cpp, U = int(node.cores_per_proc), int(node.sockets)
        elif wide:
            cpp, U = int(self._table.wide[i]["cores_per_proc"]), int(self._table.wide[i]["numa_nodes"])
        else:
            cpp, U = 0, 0
        return pack.expand_wide_placement(place, G, cpp, cpp * U, [int(req["gpus"][g]) for g in range(G)])

    def _on_commit(self, node, mapping, top) -> bool:
        """After the reference's own SetPhysicalIdsFromMapping succeeded on an attached node: the same commit on the
        device mirror, checked against what the reference just wrote into `top`.  True = the mirror is current."""
        if self._attached is None or node.name not in self._index or self._waits_for_repack(node.name) or self._off_planes(node.name):
            self._batch_ids.pop(node.name, None)
            return False
        pending = self._batch_ids.get(node.name)
        if not pending and self._deltas:
            self._flush_deltas()                               # releases queued before this commit reach the device first
            if self._waits_for_repack(node.name):
                return False
        from_batch = bool(pending)
        if pending:                                            # ScheduleBatch(apply=True) committed this placement on the device already
            ids = pending.pop(0)
            if not pending:
                del self._batch_ids[node.name]
        else:
            try:
                ids = self.CommitPlacement(node.name, top, mapping)
            except Exception:  # noqa: BLE001 - any doubt: fall back to re-packing the node
                return False
        pos = {g.device_id: k for k, g in enumerate(node.gpus)}
        want = {"groups": [{"cores": [c.core for g in pg.group_gpus for c in g.cpu_cores] + [c.core for c in pg.proc_cores],
                            "helpers": [c.core for c in pg.misc_cores], "gpus": [pos[g.device_id] for g in pg.group_gpus]}
                           for pg in top.proc_groups], "misc": [c.core for c in top.misc_cores]}
        if ids != want:
            self._batch_ids.pop(node.name, None)
            return False
        claim = set()
        for gi, pg in enumerate(top.proc_groups):
            if any(getattr(c.nic_dir, "value", c.nic_dir) in (1, 2) for c in pg.proc_cores):
                numa, idx = mapping["nic"][gi]
                claim.add(next(k for k, n in enumerate(node.nics) if n.idx == idx and n.numa_node == numa))
        self._claims[node.name] = frozenset(claim)             # the device already zeroed these NICs' capacities
        # SetBusy marked the node.  A commit made for this very call carried the node's busy time; a placement the device
        # committed earlier inside a batch carried the batch's `now` - the busy time SetBusy wrote since (time.monotonic(),
        # nhd/Node.py:843-845) then follows as a delta of its own
        carried = set() if from_batch else {"busy_time", "SetBusy"}
        left = self._reasons.get(node.name, set()) - carried
        if left:
            self._reasons[node.name] = left                    # cordon / maintenance / groups written earlier still go out as deltas
        else:
            self._dirty.pop(node.name, None)
            self._reasons.pop(node.name, None)
        return True

    def _on_claim(self, node, nidx) -> bool:
        want = self._claims.pop(node.name, None) if hasattr(self, "_claims") else None
        return want is not None and frozenset(nidx) == want

    # ---- release / reclaim / reset mirrored on the device (row f2, "K3") ---------------------
    _SCALAR_REASONS = frozenset(("active", "maintenance", "groups", "busy_time", "SetBusy", "SetGroups"))

    def _on_topology(self, node, name, top) -> bool:
        """After the reference's RemoveResourcesFromTopology / AddResourcesFromTopology ran on an attached node: the same
        change as a delta record for nhdfit_apply_deltas (sent with the next flush).  True = no re-pack needed."""
        if self._attached is None or node.name not in self._index or self._waits_for_repack(node.name) or self._off_planes(node.name):
            return False
        try:
            op = pack.DELTA_TAKE if name == "RemoveResourcesFromTopology" else pack.DELTA_GIVE
            self._deltas.append(self.packer.delta_from_topology(self._index[node.name], node, top, op))
        except Exception:  # noqa: BLE001 - any doubt (ids outside the node ...): re-pack
            return False
        return True

    def _on_queued_scalar(self, node, what: str) -> bool:
        if self._attached is None or node.name not in self._index or self._waits_for_repack(node.name) or self._off_planes(node.name):
            return False
        self._deltas.append(self.packer.delta_scalar(self._index[node.name], node, what))
        return True

    def _flush_deltas(self) -> None:
        """Queued deltas -> device.  A node whose delta comes back with a status (pods_used out of the tracked range, NIC
        state without a signature) is re-packed from its object like any other dirty node."""
        if not self._deltas:
            return
        q = [d for d in self._deltas if not self._waits_for_repack(self._names[int(d["node"])])]
        self._deltas = []
        if not q:
            return
        self.engine.set_dictionary(self.packer)            # a SetGroups may have interned a new group set
        status = self.engine.apply_deltas(np.array(q, dtype=pack.DELTA))
        self.delta_stats["applied"] += len(q)
        for d, st in zip(q, status):
            if st != pack.DELTA_OK:
                name = self._names[int(d["node"])]
                if not self._waits_for_repack(name):
                    self.delta_stats["repacked"] += 1
                self._mark(self._attached[name], "delta-status")

    def _mark(self, node, reason: str = "") -> None:
        self._dirty[node.name] = node
        self._reasons.setdefault(node.name, set()).add(reason)

    def _waits_for_repack(self, name: str) -> bool:
        """The node waits for a re-pack: its pending change is more than writes to scalar fields (those travel as deltas of their
        own and commute with commits and releases).  Asked per node: every commit of a pending list asks it, and a set over ALL
        dirty nodes per question made the scheduler's loop quadratic in the list's length (round 6: 512 pods, 109 k look-ups)."""
        return name in self._dirty and (not self._reasons.get(name, {""}) <= self._SCALAR_REASONS or self._off_planes(name))

    @property
    def _dirty_strict(self):
        """All such nodes (diagnostics and tests; the hot paths ask _waits_for_repack)."""
        return {nm for nm in self._dirty if self._waits_for_repack(nm)}

    def mark_dirty(self, name: str) -> None:
        if self._attached is not None and name in self._attached:
            self._mark(self._attached[name], "explicit")

    def _full_upload(self, nl: Dict[str, object]) -> None:
        self._last_subset = None                           # node indices change
        table = self.packer.pack_nodes(nl)
        self._table = table
        self._names = table.names
        self._index = {nm: i for i, nm in enumerate(table.names)}
        self.engine.reset_nodes()
        self.engine.set_dictionary(self.packer)
        if table.n:
            self.engine.upload(table)

    def _flush_dirty(self) -> None:
        if any(nm not in self._index for nm in self._dirty):
            self._full_upload(self._attached)          # nodes were added or removed
            self._dirty.clear()
            self._reasons.clear()
            self._deltas = []
            return
        # writes to scalar fields (cordon / maintenance / groups / busy time, nhd/NHDScheduler.py:533-570) travel as deltas too
        for name in [nm for nm in self._dirty if self._reasons.get(nm, {""}) <= self._SCALAR_REASONS and not self._off_planes(nm)]:
            node, why = self._dirty.pop(name), self._reasons.pop(name)
            i = self._index[name]
            if why & {"active", "maintenance"}:
                self._deltas.append(self.packer.delta_scalar(i, node, "active"))
            if why & {"groups", "SetGroups"}:
                self._deltas.append(self.packer.delta_scalar(i, node, "groups"))
            if why & {"busy_time", "SetBusy"}:
                self._deltas.append(self.packer.delta_scalar(i, node, "busy_time"))
        self._flush_deltas()
        if not self._dirty:
            return
        one = pack.empty_table(1)
        for name, node in self._dirty.items():
            i = self._index[name]
            self.packer.pack_node_into(node, one, 0)
            for f in ("p0", "p1", "p2", "p3", "p4", "detail", "origin"):
                getattr(self._table, f)[i] = getattr(one, f)[0]
            if self._table.wide is None:
                self._table.wide = {}
            if one.wide and 0 in one.wide:                     # a wide node: its record travels with the (placeholder) planes
                self._table.wide[i] = one.wide[0].copy()
            else:
                self._table.wide.pop(i, None)
            if self._table.share is None:
                self._table.share = {}
            if one.share and 0 in one.share:                   # ... and, under ENABLE_SHARING, its NICs' speed_used
                self._table.share[i] = one.share[0].copy()
            else:
                self._table.share.pop(i, None)
        self.engine.set_dictionary(self.packer)        # signatures may have been added
        idx = sorted(self._index[nm] for nm in self._dirty)
        lo = 0
        while lo < len(idx):                           # upload contiguous runs
            hi = lo
            while hi + 1 < len(idx) and idx[hi + 1] == idx[hi] + 1:
                hi += 1
            self.engine.upload(self._table.slice(idx[lo], idx[hi] + 1), first=idx[lo], capacity=self._table.n)
            lo = hi + 1
        self._dirty.clear()
        self._reasons.clear()

    def _candidates(self, nl: Dict[str, object]) -> Optional[np.ndarray]:
        """Bitmask [chunks] (bit = node) of the attached nodes that are in `nl`; None when nl is everything."""
        n = len(self._names)
        names = list(nl)
        if len(names) == n and names == self._names:          # C-speed comparison (identical str objects short-cut)
            return None
        # pods of one node group get the same filtered dict from InitialNodeFilter time after time: the masks of the subsets seen
        # lately are kept, and recognising one is a C-speed list comparison instead of a dictionary look-up per node
        key = (len(names), names[0], names[len(names) // 2], names[-1]) if names else (0,)
        if self._last_subset is not None:
            for cached, cached_mask in self._last_subset.get(key, ()):
                if names == cached:                            # (identical str objects short-cut: a pointer comparison per name)
                    return cached_mask
        idx = np.fromiter(map(self._index.__getitem__, names), dtype=np.int64, count=len(names))
        if len(idx) > 1 and not np.all(idx[1:] > idx[:-1]):
            raise ValueError("FindNode: `nl` must keep the relative order of the attached node dict")
        bits = np.zeros(((n + 63) // 64) * 64, dtype=bool)
        bits[idx] = True
        mask = np.ascontiguousarray(np.packbits(bits.reshape(-1, 64), axis=1, bitorder="little").view("<u8").reshape(-1))
        if self._last_subset is None or self._subset_names + len(names) > (1 << 22):   # (bounded: four million names kept, then start over)
            self._last_subset, self._subset_count, self._subset_names = {}, 0, 0
        self._last_subset.setdefault(key, []).append((names, mask))
        self._subset_count += 1
        self._subset_names += len(names)
        return mask

    # ---- the reference interface ----------------------------------------------------------
    def FindNode(self, nl: Dict[str, object], top) -> Tuple:
        return self.FindNodes(nl, [top])[0]

    def ScheduleBatch(self, nl: Dict[str, object], tops: Sequence[object],
                      pod_groups: Optional[Sequence[Sequence[str]]] = None, now: Optional[float] = None,
                      apply: bool = False) -> List[Tuple]:
        """Mode B: the result list the scheduler loop would produce by calling FindNode and committing each
        winner before the next pod (nhd/NHDScheduler.py:289-304, 425-437) - decided AND committed on the device in
        one pass (nhdfit_schedule_batch).  `self.last_placements[i]` holds pod i's physical ids (the lists
        SetPhysicalIdsFromMapping would write into its topology, pack.expand_placement) or None.
        The node objects are NOT modified: apply each placement with the node's own SetBusy /
        SetPhysicalIdsFromMapping / ClaimPodNICResources, exactly as AttemptScheduling does.  apply=True keeps the
        commits in the device mirror (attached mode: the reference mutators that follow are then recognised as
        already mirrored); apply=False restores it."""
        return self._run(nl, tops, pod_groups, now, sequential=True, apply=apply)

    def FindNodes(self, nl: Dict[str, object], tops: Sequence[object],
                  pod_groups: Optional[Sequence[Sequence[str]]] = None, now: Optional[float] = None) -> List[Tuple]:
        """Mode-A batch: every pod of `tops` against the same snapshot of `nl`.  With `pod_groups`
        the kernel applies InitialNodeFilter itself (nl = all nodes); without, `nl` is taken as already
        filtered, exactly like the argument of Matcher.FindNode."""
        return self._run(nl, tops, pod_groups, now, sequential=False)

    def FindNodesFromConfigs(self, nl: Dict[str, object], cfg_texts: Sequence[str],
                             pod_groups: Optional[Sequence[Sequence[str]]] = None, now: Optional[float] = None,
                             sequential: bool = False) -> List[Tuple]:
        """Batch matching straight from the pods' Triad libconfig texts: no
        TriadCfgParser / CfgTopology object graph per pending pod (nhd/NHDScheduler.py:262-277) - the texts are
        digested by host C++ in libnhdfit.so (nhd_amd/wire.py).  A text for which the reference's CfgToTopology
        returns None yields `(None,)`, like the scheduler skipping that pod.  The caller parses the config of a
        pod it actually binds (the commit step needs the CfgTopology, nhd/NHDScheduler.py:289-304)."""
        from . import wire
        if not cfg_texts:
            return []
        reqs, codes = wire.digest_configs(cfg_texts)
        big_reqs = {}
        for i in np.flatnonzero(codes > wire.WIRE_NONE):
            if codes[i] == wire.WIRE_LIMIT:                  # more than four processing groups: up to eight ride the general path
                big_reqs[int(i)] = wire.digest_config_big(cfg_texts[int(i)])       # (raises what the limit really is beyond that)
            else:                                            # what the reference would raise on: report it properly
                wire.digest_config(cfg_texts[int(i)])
        # a hugepage request beyond the tile's table digests without a code: such a pod is a big request too, as FindNodes(tops)
        # routes it (pack.needs_general_path) - staged with the others it would fail the whole call with NHDFIT_E_LIMIT
        for i in np.flatnonzero((codes == 0) & (reqs["hugepages_gb"] > pack.MAX_HUGEPAGES_GB)):
            big_reqs[int(i)] = wire.digest_config_big(cfg_texts[int(i)])
        skip = codes == wire.WIRE_NONE                       # all-zero (= never matching) requests
        # (the pods' group bits are filled in by _run, after the mirror - and with it the dictionary - is current)
        for i in np.flatnonzero(~skip):
            if int(i) not in big_reqs and reqs[i]["n_groups"] == 0 and len(nl):
                raise IndexError("pod without processing groups (the reference fails the same way, Matcher.py:346)")
        out = self._run(nl, None, pod_groups, now, sequential, reqs=reqs, big_reqs=big_reqs)
        return [(None,) if skip[i] else out[i] for i in range(len(cfg_texts))]

    @property
    def unmirrored(self) -> Dict[str, str]:
        """Nodes NEITHER layout holds (more than four sockets, more than 128 physical cores per socket ...; name -> reason): they never
        match, everything else is answered for.  Nodes beyond the fast layout but within the general path's (wide nodes) are not here:
        they are answered for exactly, by enumeration (`wide_nodes`)."""
        return self.packer.unmirrored

    @property
    def wide_nodes(self) -> List[str]:
        """Names of the mirrored nodes the general path serves (3-4 sockets, 65-128 physical cores per socket, ...)."""
        if self._table is None or not self._table.wide:
            return []
        return [self._names[i] for i in sorted(self._table.wide)]

    def _warn_unmirrored(self) -> None:
        for name, why in self.packer.unmirrored.items():
            if name not in self._warned:
                self._warned.add(name)
                self.logger.warning("node %s is not mirrored on the device and will never be selected: %s", name, why)

    def _run(self, nl, tops, pod_groups, now, sequential, reqs=None, apply=False, big_reqs=None):
        try:
            return self._run_checked(nl, tops, pod_groups, now, sequential, reqs, apply, big_reqs)
        except NhdFitError as e:
            if self.strict:
                raise
            n_pods = len(tops) if reqs is None else len(reqs)
            self.logger.error("FindNode: the device path failed (%s): %d pod(s) answered (None,)", e, n_pods)
            self._mirror_foreign = True                   # whatever the mirror holds now, rebuild it before the next call
            self._last_subset = None
            self.last_placements = [None] * n_pods
            return [(None,) for _ in range(n_pods)]

    def _run_checked(self, nl, tops, pod_groups, now, sequential, reqs=None, apply=False, big_reqs=None):
        if reqs is None:
            if not tops:
                self.last_placements = []
                return []
            for top in tops:
                if len(top.proc_groups) == 0 and len(nl):
                    raise IndexError("pod without processing groups (the reference fails the same way, Matcher.py:346)")
        n_pods = len(tops) if reqs is None else len(reqs)
        if len(nl) == 0:
            return [(None,) for _ in range(n_pods)]
        now = self.clock() if now is None else now
        cand = None
        known = False
        if self._attached is not None:
            if len(self._attached) != len(self._names) or self._mirror_foreign:
                self.attach(self._attached)                # nodes were added / removed, or a foreign dict was matched in between
            if nl is self._attached:                       # the scheduler's own dict: nothing to look up, O(1) in the node count
                known = True
            else:
                try:
                    cand = self._candidates(nl)
                    known = True
                except KeyError:                           # names the attached dict does not hold: stateless for this call
                    known = False
        if self._batch_ids:
            # placements of an earlier ScheduleBatch(apply=True) the caller never applied to its node objects: the mirror holds
            # commits the objects do not - those nodes are re-packed from their objects
            for name in list(self._batch_ids):
                nd = self._attached.get(name) if self._attached is not None else None
                if nd is not None:
                    self._mark(nd, "unapplied-batch")
            self._batch_ids.clear()
        if known:
            self._flush_dirty()
        else:
            self._full_upload(nl)
            self._mirror_foreign = self._attached is not None
        self._warn_unmirrored()
        if reqs is not None and self.packer.sharing:       # digested from config texts, and speed_used prices the NICs (nhd/Node.py:754)
            for arr in (reqs, *(big_reqs or {}).values()):
                for rq in (arr if arr.ndim else [arr]):
                    if int(rq["n_groups"]) == 0:
                        continue
                    why = self.packer.admit_wire_request(rq)
                    if why is not None:
                        if self.strict:
                            raise pack.UnsupportedNode(why)
                        self.logger.error("a pod of the call cannot be answered exactly and is answered (None,): %s", why)
                        rq["n_groups"] = 0                 # (an all-zero group count never matches)
        # (nhd/Node.py:20 ENABLE_SHARING = True: the packer mirrored every node for the general path, whose NIC stage prices a NIC at
        #  speed * pct - speed_used[x] as the reference does - nothing to route here: the table pass finds placeholders only)
        if reqs is None and any(pack.needs_general_path(top) for top in tops):
            # pods the table pass cannot express but the general path answers (5..8 processing groups, huge hugepage requests)
            is_big = [pack.needs_general_path(top) for top in tops]
            big_recs = []
            for p in range(n_pods):
                if not is_big[p]:
                    continue
                try:
                    big_recs.append(self.packer.digest_big(tops[p], None if pod_groups is None else pod_groups[p]))
                except pack.UnsupportedNode:               # beyond the big record too (a group of more than 255 cores ...): strict raises,
                    if self.strict:                        # otherwise the pod goes the ordinary pods' way below and is answered (None,), logged
                        raise
                    is_big[p] = False
            small_idx = [p for p in range(n_pods) if not is_big[p]]
            beyond = []
            small = self.packer.digest_many([tops[p] for p in small_idx], None if pod_groups is None else [pod_groups[p] for p in small_idx],
                                            unsupported=beyond)
            for k, why in beyond:
                self.logger.error("pod %d of the call cannot be expressed as a request record and is answered (None,): %s", small_idx[k], why)
            bigs = np.array(big_recs, dtype=pack.BIG_REQ) if big_recs else np.zeros(0, pack.BIG_REQ)
            return self._run_with_big(nl, is_big, small, bigs, now, cand, sequential, apply)
        if reqs is not None and big_reqs:                  # digested from config texts (FindNodesFromConfigs)
            is_big = [p in big_reqs for p in range(n_pods)]
            small = reqs[[p for p in range(n_pods) if not is_big[p]]]
            bigs = np.array([big_reqs[p] for p in range(n_pods) if is_big[p]], dtype=pack.BIG_REQ)
            if pod_groups is not None:
                for arr, idx in ((small, [p for p in range(n_pods) if not is_big[p]]), (bigs, [p for p in range(n_pods) if is_big[p]])):
                    for k, p in enumerate(idx):
                        arr[k]["flags"] |= pack.RF_INITIAL_FILTER
                        arr[k]["groups"] = self.packer.group_bits_known(pod_groups[p])
            return self._run_with_big(nl, is_big, small, bigs, now, cand, sequential, apply)
        if reqs is None:
            beyond: List[Tuple[int, str]] = []
            reqs = self.packer.digest_many(tops, pod_groups, unsupported=beyond)
            if n_pods == 1 and not beyond:
                self._digest_cache = (tops[0], reqs[0])     # CommitPlacement of the same topology object does not digest it again
            for i, why in beyond:
                self.logger.error("pod %d of the call cannot be expressed as a request record and is answered (None,): %s", i, why)
        elif pod_groups is not None:                       # requests digested from config texts: InitialNodeFilter in the kernel
            for i in range(n_pods):
                reqs[i]["flags"] |= pack.RF_INITIAL_FILTER
                reqs[i]["groups"] = self.packer.group_bits_known(pod_groups[i])
        places = None
        if sequential:
            self.packer.close_signatures()                 # every NIC state a commit can produce has a signature
            self.engine.set_dictionary(self.packer)
            node, maps, places, status = self.engine.schedule_batch(reqs, now, self.packer, cand=cand, apply=apply)
            if (status == pack.COMMIT_WOULD_RAISE).any():
                self.logger.warning("mode B: the reference's commit step would have failed for pod %d",
                                    int(np.flatnonzero(status == pack.COMMIT_WOULD_RAISE)[0]))
            index = (node - self.engine.global_base).tolist()
        else:
            score, _, maps = self.engine.find(reqs, now, cand=cand, want_bitmap=False, want_map=True)
            base = self.engine.global_base
            index = [winner_index(s) - base if s else -1 for s in score.tolist()]
        out: List[Tuple] = []
        self.last_placements = [None] * n_pods
        rows = pack.unpack_mappings(maps)                  # (gpu[4], cpu[5], nic_numa[4], nic_idx[4], valid) per pod as Python ints: one pass
        n_groups = reqs["n_groups"].tolist()
        names = self._names
        place_rows = pack.unpack_placements(places) if places is not None and self._attached is not None else None   # (Python ints, one C call)
        gpus_of = reqs["gpus"].tolist() if place_rows is not None else None
        for p in range(n_pods):
            i = index[p]
            if i < 0:
                out.append((None,))
                continue
            name = names[i]
            G = n_groups[p]
            if places is not None and self._attached is not None:
                nd = self._attached.get(name)
                if nd is not None and place_rows[p][55] == pack.COMMIT_WIDE:            # (status) the pod landed on a wide node
                    wp = getattr(self.engine, "last_wide_places", {}).get(p)
                    if wp is not None:
                        self.last_placements[p] = pack.expand_wide_placement(wp, G, int(nd.cores_per_proc), int(nd.cores_per_proc) * int(nd.sockets),
                                                                             gpus_of[p][:G])
                    if apply:                                  # its record is re-packed from the object before the next call, whether or not
                        self._mark(nd, "wide-batch")           # the caller applies the placement to it (no delta form on the general path)
                elif nd is not None:
                    self.last_placements[p] = pack.expand_placement(place_rows[p], G, int(nd.cores_per_proc), int(nd.cores_per_proc) * int(nd.sockets),
                                                                    gpus_of[p][:G])
                    if apply:                                  # the reference mutators that follow find their work mirrored already
                        self._batch_ids.setdefault(nd.name, []).append(self.last_placements[p])
            gpu, cpu, nic_numa, nic_idx, valid = rows[p]
            if not valid:
                raise RuntimeError(f"internal error: no mapping produced for feasible node {name}")
            out.append((name, {"gpu": gpu[:G], "cpu": cpu[:G + 1],                                          # (tuples of plain ints, as the reference's hold)
                               "nic": list(zip(nic_numa[:G], nic_idx[:G]))}))
        return out

    # ---- pods with 5..8 processing groups (nhdfit_big_req): the general path ---------------------------------------------
    def _run_with_big(self, nl, big, small_reqs, big_reqs, now, cand, sequential, apply):
        """FindNodes / ScheduleBatch for a call that holds pods the table-driven pass cannot express: more than pack.MAX_GROUPS
        processing groups (the reference takes any group count, nhd/Matcher.py:118,203,242) or a hugepage request beyond the pod
        tile's table.  `big[p]` says which; `small_reqs` / `big_reqs` are the two kinds' records in pod order.  A big pod is a
        nhdfit_big_req answered on the device by the general path (nhdfit_big_find: every node by explicit enumeration, the same
        score word, the general CPython set model for its mapping; nhdfit_big_commit for its commit step); the other pods of the
        call take the table-driven pass as always.  Mode A: two device calls, results interleaved.  Mode B: the scheduler's loop
        (nhd/NHDScheduler.py:425-437) cut at the big pods - every run of ordinary pods between two of them is ONE device batch
        (nhdfit_schedule_batch, commits left in the mirror), every big pod is decided and committed on the device before the run
        behind it is matched."""
        n_pods = len(big)
        small_idx = [p for p in range(n_pods) if not big[p]]
        big_idx = [p for p in range(n_pods) if big[p]]
        n_groups = [0] * n_pods
        for k, p in enumerate(small_idx):
            n_groups[p] = int(small_reqs[k]["n_groups"])
        for k, p in enumerate(big_idx):
            n_groups[p] = int(big_reqs[k]["n_groups"])
        base = self.engine.global_base
        names = self._names
        out: List[Tuple] = [(None,)] * n_pods
        self.last_placements = [None] * n_pods

        def answer(p, i, row):
            G = n_groups[p]
            gpu, cpu, nic_numa, nic_idx, valid = row
            if not valid:
                raise RuntimeError(f"internal error: no mapping produced for feasible node {names[i]}")
            out[p] = (names[i], {"gpu": gpu[:G], "cpu": cpu[:G + 1], "nic": list(zip(nic_numa[:G], nic_idx[:G]))})

        if not sequential:
            if small_idx:
                score, _, maps = self.engine.find(small_reqs, now, cand=cand, want_bitmap=False, want_map=True)
                rows = pack.unpack_mappings(maps)
                for k, s in enumerate(score.tolist()):
                    if s:
                        answer(small_idx[k], winner_index(s) - base, rows[k])
            score, maps = self.engine.big_find(big_reqs, now, cand=cand)
            rows = pack.unpack_big_mappings(maps)
            for k, s in enumerate(score.tolist()):
                if s:
                    answer(big_idx[k], winner_index(s) - base, rows[k])
            return out

        self.packer.close_signatures()                     # every NIC state a commit can produce has a signature
        self.engine.set_dictionary(self.packer)
        objects = self._attached if self._attached is not None else nl
        touched = []
        pos_small = {p: k for k, p in enumerate(small_idx)}
        pos_big = {p: k for k, p in enumerate(big_idx)}

        def record(p, i, ids, on_wide):
            touched.append(names[i])
            nd = objects.get(names[i])
            if nd is None:
                return
            self.last_placements[p] = ids
            if self._attached is not None and apply:
                if on_wide:
                    self._mark(nd, "wide-batch")           # a wide record is re-packed from the object before the next call (no delta form)
                else:
                    self._batch_ids.setdefault(nd.name, []).append(ids)     # the reference mutators that follow find their work mirrored already

        p = 0
        while p < n_pods:
            if big[p]:
                req = big_reqs[pos_big[p]]
                score, maps = self.engine.big_find(big_reqs[pos_big[p]:pos_big[p] + 1], now, cand=cand)
                if score[0]:
                    i = winner_index(int(score[0])) - base
                    answer(p, i, pack.unpack_big_mappings(maps)[0])
                    ids = self._commit_big(i, objects.get(names[i]), req, maps[0], now)
                    record(p, i, ids, self._table is not None and bool(self._table.wide) and i in self._table.wide)
                p += 1
                continue
            # a run of ordinary pods between two big ones: one device batch (the decision engine), commits left in the mirror
            q = p
            while q < n_pods and not big[q]:
                q += 1
            ka, kb = pos_small[p], pos_small[p] + (q - p)
            node, maps, places, status = self.engine.schedule_batch(small_reqs[ka:kb], now, self.packer, cand=cand, apply=True)
            if (status == pack.COMMIT_WOULD_RAISE).any():
                self.logger.warning("mode B: the reference's commit step would have failed for pod %d",
                                    p + int(np.flatnonzero(status == pack.COMMIT_WOULD_RAISE)[0]))
            rows = pack.unpack_mappings(maps)
            wide_places = getattr(self.engine, "last_wide_places", {})
            for j in range(q - p):
                if node[j] < 0:
                    continue
                i = int(node[j]) - base
                answer(p + j, i, rows[j])
                nd = objects.get(names[i])
                on_wide = int(places[j]["status"]) == pack.COMMIT_WIDE
                ids = None
                if nd is not None:
                    G = n_groups[p + j]
                    gp = [int(small_reqs[ka + j]["gpus"][g]) for g in range(G)]
                    cpp, n_log = int(nd.cores_per_proc), int(nd.cores_per_proc) * int(nd.sockets)
                    if on_wide:
                        wp = wide_places.get(j)
                        ids = pack.expand_wide_placement(wp, G, cpp, n_log, gp) if wp is not None else None
                    else:
                        ids = pack.expand_placement(places[j], G, cpp, n_log, gp)
                record(p + j, i, ids, on_wide)
            p = q
        if not apply:
            # the mirror goes back to what the node objects say (nobody applied anything to them): attached - those nodes are
            # re-packed before the next call; stateless - every call packs `nl` again anyway
            if self._attached is not None:
                for name in touched:
                    nd = self._attached.get(name)
                    if nd is not None:
                        self._mark(nd, "unapplied-batch")
        return out
