"""Host-side packing: reference objects -> the HBM layout of include/nhdfit.h.

* ``Packer.pack_nodes(nl)``   Dict[str, Node]  -> :class:`NodeTable` (five 16-byte SoA planes + the
  cold per-node detail record).  Nodes are read purely by attribute,
  so live ``nhd.Node.Node`` objects and the stand-ins of :mod:`workload.refmodel` both work.
* ``Packer.digest(top, pod_groups)``   CfgTopology -> one ``nhdfit_req`` record.

The packer also owns the cluster-wide interning dictionaries the kernels index into:
capacity classes (distinct f64 NIC capacities, computed here with the reference's own Python
expression ``speed * 0.9`` so the bits are identical), NIC signatures (which NIC pools a NUMA node
offers) and node-group names (bit ids of a uint64).
"""
from __future__ import annotations

import math
import struct
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

MAX_GROUPS = 4
MAX_NUMA = 2
MAX_CORES_PER_NUMA = 64
MAX_GPUS = 32
MAX_GPUS_PER_NUMA = 8
MAX_NICS_PER_NUMA = 16
MAX_SWITCHES = 14
MAX_CLASSES = 16
GLIMIT_NONE = 255
TILE = 64
MAX_HUGEPAGES_GB = 1022         # fit_core.h kMaxHpRows - 2

NF_MAINTENANCE, NF_ACTIVE, NF_SMT, NF_HAS_GPU = 1, 2, 4, 8
RF_INITIAL_FILTER = 1
RF_NIC_SPLIT = 2                  # informational (include/nhdfit.h NHDFIT_RF_NIC_SPLIT): a group with several RX / TX cores
RF_NIC_SPLIT_DYADIC = 4

NIC_BW_AVAIL_PERCENT = 0.9      # nhd/Node.py:18: the default; the packer reads the constant of the module the node objects come from
_MODULE_CONSTANTS = ("NIC_BW_AVAIL_PERCENT", "SCHEDULABLE_NIC_SPEED_THRESH_MBPS", "ENABLE_SHARING")   # nhd/Node.py:18-20
_node_constants_cache: Dict[type, dict] = {}


def node_module_constants(node) -> dict:
    """The three module-level switches of nhd/Node.py:18-20 as the module that defines the node's class has them NOW
    (an operator may have edited them): the first class along the MRO whose module carries ENABLE_SHARING speaks; stand-in
    node classes without the constants get the reference's defaults.  Looked up per call - the constants are plain module
    globals the reference reads at call time (nhd/Node.py:289-292) - through a per-class cache of WHICH module."""
    import sys
    cls = type(node)
    ent = _node_constants_cache.get(cls)
    if ent is None:
        mod = None
        for k in cls.__mro__:
            m = sys.modules.get(getattr(k, "__module__", None))
            if m is not None and hasattr(m, "ENABLE_SHARING"):
                mod = m
                break
        ent = _node_constants_cache[cls] = {"module": mod}
    mod = ent["module"]
    return {"NIC_BW_AVAIL_PERCENT": getattr(mod, "NIC_BW_AVAIL_PERCENT", NIC_BW_AVAIL_PERCENT) if mod else NIC_BW_AVAIL_PERCENT,
            "SCHEDULABLE_NIC_SPEED_THRESH_MBPS": getattr(mod, "SCHEDULABLE_NIC_SPEED_THRESH_MBPS", 11000) if mod else 11000,
            "ENABLE_SHARING": bool(getattr(mod, "ENABLE_SHARING", False)) if mod else False}

P0 = np.dtype([("t0", "<u8", (2,))])
P1 = np.dtype([("t1", "<u8", (2,))])
P2 = np.dtype([("gpu_free", "<u4"), ("gpu_numa1", "<u4"), ("hp_free", "<i4"), ("flags", "<u4")])
P3 = np.dtype([("groups", "<u8"), ("sig_numa", "<u2", (2,)), ("sig_pci", "<u2", (2,))])
P4 = np.dtype([("busy_time", "<f8"), ("group_set", "<u4"), ("reserved", "<u4")])
DETAIL = np.dtype([("nic_cnt", "u1", (2,)), ("sw_free", "u1", (MAX_SWITCHES,)),
                   ("nic_cls", "u1", (2, MAX_NICS_PER_NUMA)), ("nic_sw", "u1", (2, MAX_NICS_PER_NUMA)),
                   ("numa_nodes", "u1"), ("n_gpus", "u1"), ("nic_pods", "u1", (12,)), ("pad", "u1", (2,)),
                   ("gpu_sw", "u1", (MAX_GPUS,))])
ORIGIN = np.dtype([("t0", "<u8", (2,)), ("t1", "<u8", (2,)), ("nic_base", "u1", (2, MAX_NICS_PER_NUMA)), ("hp_total", "<i4"),
                   ("pad", "u1", (12,))])
DELTA_MAX_NICS = 15
DELTA = np.dtype([("node", "<u4"), ("op", "<u4"), ("t0", "<u8", (2,)), ("t1", "<u8", (2,)), ("gpus", "<u4"), ("hugepages_gb", "<i4"),
                  ("hp_total", "<i4"), ("flags_mask", "<u4"), ("flags_value", "<u4"), ("group_set", "<u4"), ("groups", "<u8"),
                  ("busy_time", "<f8"), ("nic_n", "u1"), ("nic", "u1", (DELTA_MAX_NICS,))])
DELTA_TAKE, DELTA_GIVE, DELTA_RESET, DELTA_SET_FLAGS, DELTA_SET_GROUPS, DELTA_SET_BUSY, DELTA_SET_HUGEPAGES = 1, 2, 3, 4, 5, 6, 7
DELTA_OK, DELTA_REPACK, DELTA_NEW_SIG = 0, 1, 2
PODS_LOST = 4                    # nic_pods bit pattern "out of range" (include/nhdfit.h)
REQ = np.dtype([("n_groups", "<u4"), ("map_type", "<u4"), ("hugepages_gb", "<i4"), ("flags", "<u4"),
                ("groups", "<u8"), ("gpus", "<u2", (4,)), ("cpu_smt", "<u2", (4,)), ("cpu_nosmt", "<u2", (4,)),
                ("misc_smt", "<u2"), ("misc_nosmt", "<u2"), ("n_proc", "u1", (4,)),
                ("rx", "<f8", (4,)), ("tx", "<f8", (4,)), ("n_help", "u1", (4,)), ("n_misc", "u1"), ("smt_bits", "u1"),
                ("misc_smt_enabled", "u1"), ("nic_use", "u1")])
MAPPING = np.dtype([("gpu", "i1", (4,)), ("cpu", "i1", (5,)), ("nic_numa", "i1", (4,)), ("nic_idx", "i1", (4,)),
                    ("valid", "i1"), ("pad", "i1", (2,))])
PLACEMENT = np.dtype([("proc_take", "<u8", (4,)), ("proc_pair", "<u8", (4,)), ("help_take", "<u8", (4,)), ("help_pair", "<u8", (4,)),
                      ("misc_take", "<u8"), ("misc_pair", "<u8"), ("gpu", "u1", (4, 8)), ("numa", "i1", (5,)), ("status", "u1"),
                      ("pad", "u1", (2,)), ("proc_late", "<u8", (4,)), ("help_late", "<u8", (4,)), ("misc_late", "<u8")])
COMMIT_OK, COMMIT_WOULD_RAISE, COMMIT_NEW_SIG, COMMIT_WIDE = 0, 1, 2, 3
# nodes beyond the fast layout: one self-contained record each (include/nhdfit.h nhdfit_wide_node)
WIDE_MAX_NUMA, WIDE_CORE_WORDS, WIDE_MAX_CORES_PER_NUMA = 4, 8, 128
WIDE = np.dtype([("t0", "<u8", (WIDE_CORE_WORDS,)), ("t1", "<u8", (WIDE_CORE_WORDS,)), ("o0", "<u8", (WIDE_CORE_WORDS,)), ("o1", "<u8", (WIDE_CORE_WORDS,)),
                 ("groups", "<u8"), ("busy_time", "<f8"), ("gpu_free", "<u4"), ("flags", "<u4"), ("hp_free", "<i4"), ("hp_total", "<i4"),
                 ("index", "<u4"), ("cores_per_proc", "<u2"), ("numa_nodes", "u1"), ("n_gpus", "u1"), ("nic_cnt", "u1", (WIDE_MAX_NUMA,)),
                 ("gpu_numa", "u1", (MAX_GPUS,)), ("gpu_sw", "u1", (MAX_GPUS,)),
                 ("nic_cls", "u1", (WIDE_MAX_NUMA, MAX_NICS_PER_NUMA)), ("nic_base", "u1", (WIDE_MAX_NUMA, MAX_NICS_PER_NUMA)),
                 ("nic_sw", "u1", (WIDE_MAX_NUMA, MAX_NICS_PER_NUMA)), ("nic_pods", "i1", (WIDE_MAX_NUMA, MAX_NICS_PER_NUMA)), ("pad", "u1", (20,))])
WIDE_SHARE = np.dtype([("used", "<f8", (WIDE_MAX_NUMA, MAX_NICS_PER_NUMA, 2))])    # nhdfit_wide_share: Node.nics[].speed_used per (numa, idx, rx / tx)
WIDE_PLACEMENT = np.dtype([("proc_take", "<u8", (4, 2)), ("proc_pair", "<u8", (4, 2)), ("proc_late", "<u8", (4, 2)),
                           ("help_take", "<u8", (4, 2)), ("help_pair", "<u8", (4, 2)), ("help_late", "<u8", (4, 2)),
                           ("misc_take", "<u8", (2,)), ("misc_pair", "<u8", (2,)), ("misc_late", "<u8", (2,)),
                           ("gpu", "u1", (4, 8)), ("numa", "i1", (5,)), ("status", "u1"), ("pad", "u1", (2,)), ("pod", "<u4"), ("node", "<u4")])
assert WIDE.itemsize == 640 and WIDE_PLACEMENT.itemsize == 480
# pods with 5..8 processing groups: the general path's request / mapping / placement records (include/nhdfit.h nhdfit_big_*)
BIG_MAX_GROUPS, BIG_MAX_TUPLES = 8, 262144
BIG_REQ = np.dtype([("n_groups", "<u4"), ("map_type", "<u4"), ("hugepages_gb", "<i4"), ("flags", "<u4"), ("groups", "<u8"),
                    ("gpus", "<u2", (8,)), ("cpu_smt", "<u2", (8,)), ("cpu_nosmt", "<u2", (8,)), ("misc_smt", "<u2"), ("misc_nosmt", "<u2"),
                    ("smt_bits", "<u2"), ("n_misc", "u1"), ("misc_smt_enabled", "u1"), ("rx", "<f8", (8,)), ("tx", "<f8", (8,)),
                    ("n_proc", "u1", (8,)), ("n_help", "u1", (8,)), ("nic_use", "u1"), ("pad", "u1", (31,))])
BIG_MAPPING = np.dtype([("gpu", "i1", (8,)), ("cpu", "i1", (9,)), ("nic_numa", "i1", (8,)), ("nic_idx", "i1", (8,)), ("valid", "i1"), ("pad", "i1", (2,))])
BIG_PLACEMENT = np.dtype([("proc_take", "<u8", (8, 2)), ("proc_pair", "<u8", (8, 2)), ("proc_late", "<u8", (8, 2)),
                          ("help_take", "<u8", (8, 2)), ("help_pair", "<u8", (8, 2)), ("help_late", "<u8", (8, 2)),
                          ("misc_take", "<u8", (2,)), ("misc_pair", "<u8", (2,)), ("misc_late", "<u8", (2,)),
                          ("gpu", "u1", (8, 8)), ("numa", "i1", (9,)), ("status", "u1"), ("pad", "u1", (2,)), ("pod", "<u4"), ("node", "<u4"),
                          ("pad2", "u1", (4,))])
_BIG_REQ_STRUCT = struct.Struct("<IIiIQ8H8H8HHHHBB8d8d8B8BB31x")
assert BIG_REQ.itemsize == 256 and _BIG_REQ_STRUCT.size == 256 and BIG_MAPPING.itemsize == 36 and BIG_PLACEMENT.itemsize == 904
CC = np.dtype([("cls", "u1"), ("cnt", "u1")])
assert (P0.itemsize, P1.itemsize, P2.itemsize, P3.itemsize, P4.itemsize) == (16,) * 5
_REQ_STRUCT = struct.Struct("<IIiIQ4H4H4HHH4B4d4d4BBBBB")          # REQ, field by field (digest_many packs records with it)
assert _REQ_STRUCT.size == REQ.itemsize
assert DETAIL.itemsize == 128 and REQ.itemsize == 128 and MAPPING.itemsize == 20 and PLACEMENT.itemsize == 256
assert ORIGIN.itemsize == 80 and DELTA.itemsize == 96

ALL_ONES = np.uint64(0xFFFFFFFFFFFFFFFF)


def _enum_value(x):
    """x.value, through the member's plain attribute when x is an Enum (its .value is a descriptor call)."""
    try:
        return x._value_
    except AttributeError:
        return x.value


def needs_general_path(top) -> bool:
    """The pod cannot ride the table-driven pass but the general path answers it exactly (nhdfit_big_req): 5..8 processing
    groups, or a hugepage request beyond the pod tile's hugepage table (the general path compares the integers themselves,
    nhd/Matcher.py:78)."""
    G = len(top.proc_groups)
    if G > BIG_MAX_GROUPS or G < 1:
        return False
    if G > MAX_GROUPS:
        return True
    try:
        return int(top.hugepages_gb) > MAX_HUGEPAGES_GB
    except (TypeError, ValueError):
        return False


class UnsupportedNode(ValueError):
    """The node's topology exceeds a compile-time capacity of the device layout (include/nhdfit.h)."""


class SharingEnabled(UnsupportedNode):
    """nhd/Node.py:20 ENABLE_SHARING is True in the module the node objects come from: GetFreeNumaNicResources then prices a
    NIC at speed * pct - speed_used[x] (nhd/Node.py:290) - a per-direction remainder the packed capacity classes of the fast
    layout do not model.  Such a node is mirrored for the general path only (a wide record + its nhdfit_wide_share record of the
    NICs' speed_used: round 5), where that subtraction is done in f64 as the reference writes it."""


@dataclass
class NodeTable:
    names: List[str]
    p0: np.ndarray
    p1: np.ndarray
    p2: np.ndarray
    p3: np.ndarray
    p4: np.ndarray
    detail: np.ndarray
    origin: Optional[np.ndarray] = None          # nhdfit_origin records (what ResetResources / a released NIC go back to)
    wide: Optional[Dict[int, np.ndarray]] = None  # index -> nhdfit_wide_node record of the nodes beyond the fast layout (their planes hold a placeholder)
    share: Optional[Dict[int, np.ndarray]] = None  # index -> nhdfit_wide_share record (ENABLE_SHARING = True: every node is a wide node and has one)

    @property
    def n(self) -> int:
        return len(self.p0)

    def slice(self, lo: int, hi: int) -> "NodeTable":
        return NodeTable(self.names[lo:hi] if self.names else [], self.p0[lo:hi], self.p1[lo:hi], self.p2[lo:hi],
                         self.p3[lo:hi], self.p4[lo:hi], self.detail[lo:hi],
                         None if self.origin is None else self.origin[lo:hi],
                         None if not self.wide else {i - lo: r for i, r in self.wide.items() if lo <= i < hi},
                         None if not self.share else {i - lo: r for i, r in self.share.items() if lo <= i < hi})

    def wide_records(self, first: int = 0) -> np.ndarray:
        """The wide records of this table in ascending order, `index` = first + position in the table (what nhdfit_wide_upload takes)."""
        idx = sorted(self.wide) if self.wide else []
        out = np.zeros(len(idx), WIDE)
        for k, i in enumerate(idx):
            out[k] = self.wide[i]
            out[k]["index"] = first + i
        return out

    def share_records(self) -> Optional[np.ndarray]:
        """The nhdfit_wide_share records in the order of wide_records(), or None when the cluster does not share NICs."""
        if not self.share:
            return None
        idx = sorted(self.wide) if self.wide else []
        out = np.zeros(len(idx), WIDE_SHARE)
        for k, i in enumerate(idx):
            out[k] = self.share[i]
        return out


def empty_table(n: int) -> NodeTable:
    t = NodeTable([], np.zeros(n, P0), np.zeros(n, P1), np.zeros(n, P2), np.zeros(n, P3), np.zeros(n, P4),
                  np.zeros(n, DETAIL), np.zeros(n, ORIGIN), {})
    return t


def pods_code(pods_used: int) -> int:
    """Three-bit counter of Node.nics[].pods_used in the detail record (include/nhdfit.h: two's complement, 4 = out of range)."""
    return int(pods_used) & 7 if -3 <= pods_used <= 3 else PODS_LOST


def set_pods(det, u: int, k: int, code: int) -> None:
    bit = 3 * (u * MAX_NICS_PER_NUMA + k)
    word = int.from_bytes(bytes(det["nic_pods"]), "little")
    word = (word & ~(7 << bit)) | ((code & 7) << bit)
    det["nic_pods"] = np.frombuffer(word.to_bytes(12, "little"), np.uint8)


def get_pods(det, u: int, k: int) -> int:
    return (int.from_bytes(bytes(det["nic_pods"]), "little") >> (3 * (u * MAX_NICS_PER_NUMA + k))) & 7


def _dyadic_speed(v) -> bool:
    """include/nhdfit.h NHDFIT_RF_NIC_SPLIT_DYADIC: a non-negative multiple of 2^-20 below 2^31 (wire_digest.cpp applies the same test)"""
    try:
        v = float(v)
    except (TypeError, ValueError):
        return False
    return 0.0 <= v < 2147483648.0 and (v * 1048576.0).is_integer()


class Packer:
    def __init__(self, strict: bool = False):
        """strict=False (default): a node whose shape exceeds a capacity of the device layout (include/nhdfit.h: more than 2
        NUMA nodes, more than 64 physical cores per socket, ...) is mirrored as a node that never matches - FindNode keeps
        answering for every other node, as its contract demands ("never an exception") - and is listed
        in `unmirrored` (name -> reason).  strict=True raises UnsupportedNode instead."""
        self.strict = strict
        self.unmirrored: Dict[str, str] = {}
        self.caps: List[float] = [0.0]                 # class 0 = a claimed NIC (capacity 0, nhd/Node.py:292)
        self._cap_index: Dict[float, int] = {0.0: 0}
        self.sigs: List[tuple] = [()]                  # sig 0 = no NIC pool at all
        self._sig_index: Dict[tuple, int] = {(): 0}
        self.group_names: List[str] = []
        self._group_index: Dict[str, int] = {}
        self.group_sets: List[int] = []                # distinct node-group bit sets, id = position
        self._group_set_index: Dict[int, int] = {}
        self.max_gpus_per_numa = 0                     # table dimensions (fit_core.h Layout)
        self.max_cores_per_numa = 1
        self.dict_version = 0                          # bumped whenever caps / sigs / that maximum grow
        self.sharing: Optional[str] = None             # set (to the reason) when a node's module has ENABLE_SHARING = True
        # ENABLE_SHARING: the commit step adds every RX / TX core's speed to speed_used one after the other (nhd/Node.py:754) while a
        # request record carries the group's sums.  The two are the same f64 value whatever the accumulator holds as long as EVERY
        # partial sum is exact - which is the case while every value that ever reaches a speed_used (the nodes' own, every digested
        # pod's core speeds) is a non-negative multiple of 2^-20 below 2^31: a NIC's speed_used then never leaves [0, 2^32) - a pod is
        # only committed where its demands still fit under speed * 0.9 (nhd/Matcher.py:262-267; a NIC already above its capacity
        # takes its whole node out) - and sums of such values below 2^33 are exact (53-bit significands).  `share_exact` says the
        # mirror is in that regime (Gb/s figures are integers or halves: it is, in practice); once a value breaks it, a group with
        # several RX (or TX) cores is turned away again until the next full re-pack.
        self.share_exact = True
        self.nic_pct = NIC_BW_AVAIL_PERCENT            # NIC_BW_AVAIL_PERCENT of the nodes' module as of the last pack
        self._closed_upto = 0                          # close_signatures: sigs[:_closed_upto] have their successors interned

    # ---- interning ------------------------------------------------------------------------
    def cap_class(self, cap) -> int:
        cap = float(cap)
        k = self._cap_index.get(cap)
        if k is None:
            if len(self.caps) >= MAX_CLASSES:
                raise UnsupportedNode("more than %d distinct NIC capacities in the cluster" % MAX_CLASSES)
            k = len(self.caps)
            self.caps.append(cap)
            self._cap_index[cap] = k
            self.dict_version += 1
        return k

    def sig_id(self, pools: Iterable[Tuple[int, Tuple[Tuple[int, int], ...]]]) -> int:
        key = tuple(sorted(pools))
        k = self._sig_index.get(key)
        if k is None:
            if len(self.sigs) >= 0xFFFF:
                raise UnsupportedNode("NIC signature dictionary overflow")
            k = len(self.sigs)
            self.sigs.append(key)
            self._sig_index[key] = k
            self.dict_version += 1
        return k

    def group_bits(self, names: Iterable[str]) -> int:
        bits = 0
        for nm in names:
            k = self._group_index.get(nm)
            if k is None:
                if len(self.group_names) >= 64:
                    raise UnsupportedNode("more than 64 distinct node-group names")
                k = len(self.group_names)
                self.group_names.append(nm)
                self._group_index[nm] = k
            bits |= 1 << k
        return bits

    def group_bits_known(self, names: Iterable[str]) -> int:
        """Bit set of the names the NODES of the cluster carry; a pod's group name no node has contributes no bit (it can
        match nothing) and - unlike group_bits - is not interned: pod annotations are user input and must not be able
        to use up the 64 ids."""
        bits = 0
        for nm in names:
            k = self._group_index.get(nm)
            if k is not None:
                bits |= 1 << k
        return bits

    def group_set_id(self, bits: int) -> int:
        k = self._group_set_index.get(bits)
        if k is None:
            k = len(self.group_sets)
            self.group_sets.append(bits)
            self._group_set_index[bits] = k
            self.dict_version += 1
        return k

    def group_set_array(self) -> np.ndarray:
        return np.asarray(self.group_sets if self.group_sets else [0], dtype="<u8")

    def dictionary_arrays(self):
        """CSR form for nhdfit_set_dictionary."""
        sig_off, pool_off, glimit, cc = [0], [0], [], []
        for sig in self.sigs:
            for (gl, pairs) in sig:
                glimit.append(gl)
                cc.extend(pairs)
                pool_off.append(len(cc))
            sig_off.append(len(glimit))
        caps = np.asarray(self.caps if self.caps else [0.0], dtype="<f8")
        ccarr = np.zeros(max(1, len(cc)), CC)
        for i, (c, n) in enumerate(cc):
            ccarr[i] = (c, n)
        return (caps, np.asarray(sig_off, "<u4"), np.asarray(pool_off, "<u4"),
                np.asarray(glimit if glimit else [0], "u1"), ccarr, len(self.caps), len(self.sigs), len(glimit), len(cc))

    def sigs_from_detail(self, det) -> Tuple[List[int], List[int]]:
        """(sig_numa[2], sig_pci[2]) of a node from its detail record alone (NIC classes / switches, free GPUs per
        switch) - interning what is new.  Used after a device-side commit left a node in a NIC state the dictionary
        did not hold yet (NHDFIT_COMMIT_NEW_SIG); same construction as pack_node_into."""
        sig_numa, sig_pci = [0, 0], [0, 0]
        for u in range(MAX_NUMA):
            n = int(det["nic_cnt"][u])
            numa_pool: Dict[int, int] = {}
            pci_pool: Dict[int, Dict[int, int]] = {}
            for k in range(n):
                cls, sw = int(det["nic_cls"][u][k]), int(det["nic_sw"][u][k])
                numa_pool[cls] = numa_pool.get(cls, 0) + 1
                d = pci_pool.setdefault(sw, {})
                d[cls] = d.get(cls, 0) + 1
            pairs = lambda d: tuple(sorted((c, min(m, MAX_GROUPS)) for c, m in d.items()))   # noqa: E731
            if numa_pool:
                sig_numa[u] = self.sig_id([(GLIMIT_NONE, pairs(numa_pool))])
            pools = []
            for sw, d in pci_pool.items():
                gl = min(int(det["sw_free"][sw]), MAX_GROUPS)
                if gl > 0:
                    pools.append((gl, pairs(d)))
            sig_pci[u] = self.sig_id(pools)
        return sig_numa, sig_pci

    def close_signatures(self, max_new: int = 4096) -> int:
        """Interns every NIC signature a commit can turn an interned one into: one NIC of a pool claimed (its capacity
        class becomes 0, nhd/Node.py:644-646) or one GPU behind a PCI-mode pool taken (nhd/Node.py:648-655).  Counts are
        capped at MAX_GROUPS in a signature, so a capped count yields both "still capped" and "one less".  With the
        closure interned up front, device-side commits (nhdfit_schedule_batch / nhdfit_commit) never meet a NIC state
        without a signature.  Returns the number of signatures added."""
        added = 0
        todo = list(self.sigs[self._closed_upto:])        # signatures interned since the last closure (successors of the
        seen = self._sig_index                             # older ones are in the dictionary already)
        while todo:
            if added >= max_new:                           # not closed: a commit may still meet an unknown NIC state, which the
                import logging                             # device reports (NHDFIT_COMMIT_NEW_SIG) and the engine interns on the way
                logging.getLogger(__name__).warning("NIC signature closure stopped after %d new signatures; the rest is interned on demand", added)
                return added
            sig = todo.pop()
            for pi, (gl, pairs) in enumerate(sig):
                succ_pools = []
                counts = dict(pairs)
                for cls, cnt in pairs:                                  # claim one NIC of class cls
                    if cls == 0 or cnt == 0:
                        continue
                    for new_cnt in ({cnt - 1, cnt} if cnt >= MAX_GROUPS else {cnt - 1}):
                        c2 = dict(counts)
                        if new_cnt:
                            c2[cls] = new_cnt
                        else:
                            c2.pop(cls)
                        c2[0] = min(c2.get(0, 0) + 1, MAX_GROUPS)
                        succ_pools.append((gl, tuple(sorted(c2.items()))))
                if gl != GLIMIT_NONE:                                   # one GPU behind this switch taken
                    for new_gl in ({gl - 1, gl} if gl >= MAX_GROUPS else {gl - 1}):
                        succ_pools.append((new_gl, pairs) if new_gl > 0 else None)
                for sp in succ_pools:
                    pools = [p for k, p in enumerate(sig) if k != pi] + ([sp] if sp is not None else [])
                    key = tuple(sorted(pools))
                    if key not in seen:
                        self.sig_id(pools)
                        todo.append(key)
                        added += 1
        self._closed_upto = len(self.sigs)
        return added

    # ---- node side ------------------------------------------------------------------------
    def pack_node_into(self, node, t: NodeTable, i: int) -> None:
        """Node object -> record i of the table.  See __init__ for nodes the layout cannot hold."""
        if t.wide is None:
            t.wide = {}
        if t.share is None:
            t.share = {}
        try:
            self._pack_node_into(node, t, i)
            self.unmirrored.pop(node.name, None)
            t.wide.pop(i, None)
            t.share.pop(i, None)
            return
        except UnsupportedNode as e:                          # (SharingEnabled among them: the general path's arithmetic)
            why = str(e)
        # beyond the fast layout: a placeholder in the planes (never matches the table pass, indices stay what they are) ...
        for f in ("p0", "p1", "p2", "p3", "p4", "detail"):
            getattr(t, f)[i] = np.zeros((), getattr(t, f).dtype)
        if t.origin is not None:
            t.origin[i] = np.zeros((), ORIGIN)
        t.p2[i]["flags"] = NF_MAINTENANCE                      # GX row 0: never feasible (nhd/Matcher.py:71)
        t.detail[i]["numa_nodes"] = 1
        # ... and, where the general path can hold it (<= 4 sockets of <= 128 physical cores), its own record
        try:
            t.wide[i] = self.pack_wide(node)
            if self.sharing:
                t.share[i] = self.pack_share(node)
            else:
                t.share.pop(i, None)
            self.unmirrored.pop(node.name, None)
        except UnsupportedNode as e2:
            t.wide.pop(i, None)
            t.share.pop(i, None)
            if self.strict:
                raise UnsupportedNode(f"{why}; and not as a wide node either: {e2}") from None
            self.unmirrored[node.name] = f"{why}; and not as a wide node either: {e2}"

    def _pack_node_into(self, node, t: NodeTable, i: int) -> None:
        consts = node_module_constants(node)
        if consts["ENABLE_SHARING"]:
            self.sharing = (f"ENABLE_SHARING is True in the module of {type(node).__name__} (nhd/Node.py:20): NIC capacities follow "
                            "speed_used (nhd/Node.py:290) - every node is mirrored for the general path")
            self.nic_pct = consts["NIC_BW_AVAIL_PERCENT"]
            raise SharingEnabled(self.sharing)
        self.sharing = None
        pct = self.nic_pct = consts["NIC_BW_AVAIL_PERCENT"]
        U = int(node.numa_nodes)
        if U < 1 or U > MAX_NUMA:
            raise UnsupportedNode(f"node {node.name}: {U} NUMA nodes (supported: 1..{MAX_NUMA})")
        cpp = int(node.cores_per_proc)
        if cpp > MAX_CORES_PER_NUMA:
            raise UnsupportedNode(f"node {node.name}: {cpp} physical cores per socket (> {MAX_CORES_PER_NUMA})")
        if cpp > self.max_cores_per_numa:
            self.max_cores_per_numa = cpp
            self.dict_version += 1
        smt = bool(node.smt_enabled)
        cores = node.cores
        t0 = [0, 0]
        t1 = [0, 0]
        for u in range(U):
            m0 = m1 = 0
            base = u * cpp
            for b in range(cpp):
                c = cores[base + b]
                if c.socket != u:
                    raise UnsupportedNode(f"node {node.name}: core {base + b} is on socket {c.socket}, expected {u}")
                if not c.used:
                    m0 |= 1 << b
                if smt and not cores[c.sibling].used:
                    m1 |= 1 << b
            t0[u] = m0
            t1[u] = m1 if smt else 0xFFFFFFFFFFFFFFFF
        t.p0[i] = (t0,)
        t.p1[i] = (t1,)
        want_origin = t.origin is not None
        if want_origin:                                    # ResetResources: every core outside reserved_cores unused (Node.py:147-149)
            full = (1 << cpp) - 1
            o0 = [full if u < U else 0 for u in range(2)]
            o1 = [(full if smt else 0xFFFFFFFFFFFFFFFF) if u < U else 0 for u in range(2)]
            for r in getattr(node, "reserved_cores", ()):
                r = int(r)
                if 0 <= r < U * cpp:                       # a thread-0 core
                    o0[r // cpp] &= ~(1 << (r % cpp))
                elif smt and r < len(cores):               # a sibling: the thread-1 bit of the core it belongs to
                    h = int(cores[r].sibling)
                    if 0 <= h < U * cpp:
                        o1[h // cpp] &= ~(1 << (h % cpp))
            nic_base = [[0] * MAX_NICS_PER_NUMA, [0] * MAX_NICS_PER_NUMA]

        gpus = node.gpus
        if len(gpus) > MAX_GPUS:
            raise UnsupportedNode(f"node {node.name}: {len(gpus)} GPUs (> {MAX_GPUS})")
        sw_local: Dict[int, int] = {}

        def local_sw(sw) -> int:
            k = sw_local.get(sw)
            if k is None:
                if len(sw_local) >= MAX_SWITCHES:
                    raise UnsupportedNode(f"node {node.name}: more than {MAX_SWITCHES} PCIe switches")
                k = sw_local[sw] = len(sw_local)
            return k

        gfree = gn1 = 0
        per_numa = [0, 0]
        sw_free = [0] * MAX_SWITCHES
        gpu_sw = [0] * MAX_GPUS
        for g, gpu in enumerate(gpus):
            if gpu.numa_node >= U or gpu.numa_node < 0:
                raise UnsupportedNode(f"node {node.name}: GPU on NUMA node {gpu.numa_node}")
            per_numa[gpu.numa_node] += 1
            s = local_sw(gpu.pciesw)
            gpu_sw[g] = s
            if gpu.numa_node == 1:
                gn1 |= 1 << g
            if not gpu.used:
                gfree |= 1 << g
                sw_free[s] += 1
        if max(per_numa) > MAX_GPUS_PER_NUMA:
            raise UnsupportedNode(f"node {node.name}: more than {MAX_GPUS_PER_NUMA} GPUs on one NUMA node")
        if max(per_numa) > self.max_gpus_per_numa:
            self.max_gpus_per_numa = max(per_numa)
            self.dict_version += 1

        # NICs: capacity class + switch per (numa, idx);  nhd/Node.py:283-296, 275-281
        cnt = [0, 0]
        sw_numa: Dict[int, int] = {}
        numa_pool = [dict(), dict()]                       # cls -> count
        pci_pool = [dict(), dict()]                        # local switch -> {cls -> count}
        nic_cls = [[0] * MAX_NICS_PER_NUMA, [0] * MAX_NICS_PER_NUMA]
        nic_sw = [[0] * MAX_NICS_PER_NUMA, [0] * MAX_NICS_PER_NUMA]
        pods_word = 0                                      # 32 three-bit counters (include/nhdfit.h)
        for nic in node.nics:
            u = nic.numa_node
            if u < 0:
                # ninfo[n.numa_node] (nhd/Node.py:289-294) is Python indexing: -1 counts the NIC on the LAST NUMA node, while
                # GetNicObjFromIndex (Node.py:657-661) compares the label's value itself - a node the reference answers for
                # inconsistently is not mirrored (listed in HipMatcher.unmirrored) rather than answered differently
                raise UnsupportedNode(f"node {node.name}: NIC with NUMA node {u}")
            if u >= U:
                continue                                   # invisible to the NIC stage (IndexError caught, Node.py:293-294)
            k = cnt[u]
            if k >= MAX_NICS_PER_NUMA:
                raise UnsupportedNode(f"node {node.name}: more than {MAX_NICS_PER_NUMA} NICs on NUMA {u}")
            if nic.idx != k:
                raise UnsupportedNode(f"node {node.name}: NIC ordinal {nic.idx} != position {k} on NUMA {u}")
            full_cap = nic.speed * pct                     # the reference's own expression (nhd/Node.py:292), its module's constant
            cls = self.cap_class(0 if nic.pods_used > 0 else full_cap)
            s = local_sw(nic.pciesw)
            if sw_numa.setdefault(s, u) != u:
                raise UnsupportedNode(f"node {node.name}: PCIe switch {nic.pciesw:#x} has NICs on both NUMA nodes")
            nic_cls[u][k] = cls
            nic_sw[u][k] = s
            if nic.pods_used:
                pods_word |= pods_code(nic.pods_used) << (3 * (u * MAX_NICS_PER_NUMA + k))
            if want_origin:
                nic_base[u][k] = self.cap_class(full_cap)
            numa_pool[u][cls] = numa_pool[u].get(cls, 0) + 1
            d = pci_pool[u].setdefault(s, {})
            d[cls] = d.get(cls, 0) + 1
            cnt[u] = k + 1
        t.detail[i] = (cnt, sw_free, nic_cls, nic_sw, U, len(gpus), list(pods_word.to_bytes(12, "little")), [0, 0], gpu_sw)
        if want_origin:
            t.origin[i] = (o0, o1, nic_base, max(-2 ** 31, min(2 ** 31 - 1, int(getattr(node.mem, "ttl_hugepages_gb", 0)))), [0] * 12)

        def pairs(d):
            return tuple(sorted((c, min(n, MAX_GROUPS)) for c, n in d.items()))

        sig_numa, sig_pci = [0, 0], [0, 0]
        for u in range(U):
            if numa_pool[u]:
                sig_numa[u] = self.sig_id([(GLIMIT_NONE, pairs(numa_pool[u]))])
            pools = []
            for s, d in pci_pool[u].items():
                gl = min(sw_free[s], MAX_GROUPS)
                if gl > 0:
                    pools.append((gl, pairs(d)))
            sig_pci[u] = self.sig_id(pools)

        flags = (NF_MAINTENANCE if node.maintenance else 0) | (NF_ACTIVE if node.active else 0) | \
                (NF_SMT if smt else 0) | (NF_HAS_GPU if len(gpus) > 0 else 0)
        hp = int(node.mem.free_hugepages_gb)
        t.p2[i] = (gfree, gn1, max(-2 ** 31, min(2 ** 31 - 1, hp)), flags)
        gbits = self.group_bits(node.groups)
        t.p3[i] = (gbits, sig_numa, sig_pci)
        t.p4[i] = (float(node.busy_time), self.group_set_id(gbits), 0)

    def pack_wide(self, node) -> np.ndarray:
        """Node object -> nhdfit_wide_node (include/nhdfit.h): the general path's record of a node the five planes cannot hold
        (3 or 4 sockets, 65..128 physical cores per socket, more than 8 GPUs on a NUMA node, a switch with NICs on two NUMA
        nodes).  Read by attribute like _pack_node_into; `index` is filled in at upload."""
        consts = node_module_constants(node)
        pct = consts["NIC_BW_AVAIL_PERCENT"]
        U, cpp = int(node.numa_nodes), int(node.cores_per_proc)
        if U < 1 or U > WIDE_MAX_NUMA or int(node.sockets) != U:
            raise UnsupportedNode(f"node {node.name}: {U} NUMA nodes on {int(node.sockets)} sockets (general path: 1..{WIDE_MAX_NUMA}, one per socket)")
        if cpp < 1 or cpp > WIDE_MAX_CORES_PER_NUMA:
            raise UnsupportedNode(f"node {node.name}: {cpp} physical cores per socket (general path: <= {WIDE_MAX_CORES_PER_NUMA})")
        smt = bool(node.smt_enabled)
        cores = node.cores
        n_phys = U * cpp
        if len(cores) != (2 * n_phys if smt else n_phys):
            raise UnsupportedNode(f"node {node.name}: {len(cores)} logical cores do not divide into {U} sockets of {cpp}")
        w = np.zeros((), WIDE)
        t0 = t1 = 0
        for c in range(n_phys):
            core = cores[c]
            if core.socket != c // cpp:
                raise UnsupportedNode(f"node {node.name}: core {c} is on socket {core.socket}, expected {c // cpp}")
            if smt and core.sibling != c + n_phys:
                raise UnsupportedNode(f"node {node.name}: sibling of core {c} is {core.sibling}, expected {c + n_phys}")
            if not core.used:
                t0 |= 1 << c
            if smt and not cores[core.sibling].used:
                t1 |= 1 << c
        full = (1 << n_phys) - 1
        if not smt:
            t1 = (1 << 512) - 1
        o0, o1 = full, (full if smt else (1 << 512) - 1)
        for r in getattr(node, "reserved_cores", ()):
            r = int(r)
            if 0 <= r < n_phys:
                o0 &= ~(1 << r)
            elif smt and r < len(cores):
                o1 &= ~(1 << (r - n_phys))

        def words(x):
            return [(x >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(WIDE_CORE_WORDS)]
        w["t0"], w["t1"], w["o0"], w["o1"] = words(t0), words(t1), words(o0), words(o1)
        gpus = node.gpus
        if len(gpus) > MAX_GPUS:
            raise UnsupportedNode(f"node {node.name}: {len(gpus)} GPUs (> {MAX_GPUS})")
        sw_local: Dict[int, int] = {}

        def local_sw(sw) -> int:
            k = sw_local.get(sw)
            if k is None:
                if len(sw_local) >= 255:
                    raise UnsupportedNode(f"node {node.name}: more than 255 PCIe switches")
                k = sw_local[sw] = len(sw_local)
            return k
        gfree = 0
        for g, gpu in enumerate(gpus):
            if not (0 <= gpu.numa_node < U):
                raise UnsupportedNode(f"node {node.name}: GPU on NUMA node {gpu.numa_node}")
            w["gpu_numa"][g] = gpu.numa_node
            w["gpu_sw"][g] = local_sw(gpu.pciesw)
            if not gpu.used:
                gfree |= 1 << g
        cnt = [0] * WIDE_MAX_NUMA
        for nic in node.nics:
            u = nic.numa_node
            if u < 0:
                # ninfo[n.numa_node] (nhd/Node.py:289-294) is Python indexing: -1 counts the NIC on the LAST NUMA node, while
                # GetNicObjFromIndex (Node.py:657-661) compares the label's value itself - a node the reference answers for
                # inconsistently is not mirrored (listed in HipMatcher.unmirrored) rather than answered differently
                raise UnsupportedNode(f"node {node.name}: NIC with NUMA node {u}")
            if u >= U:
                continue                                   # invisible to the NIC stage (IndexError caught, Node.py:293-294)
            k = cnt[u]
            if k >= MAX_NICS_PER_NUMA:
                raise UnsupportedNode(f"node {node.name}: more than {MAX_NICS_PER_NUMA} NICs on NUMA {u}")
            if nic.idx != k:
                raise UnsupportedNode(f"node {node.name}: NIC ordinal {nic.idx} != position {k} on NUMA {u}")
            full_cap = nic.speed * pct                     # the reference's own expression (nhd/Node.py:292)
            w["nic_base"][u][k] = self.cap_class(full_cap)
            w["nic_cls"][u][k] = self.cap_class(0 if nic.pods_used > 0 else full_cap)
            w["nic_sw"][u][k] = local_sw(nic.pciesw)
            w["nic_pods"][u][k] = max(-127, min(127, int(nic.pods_used)))
            cnt[u] = k + 1
        w["nic_cnt"] = cnt
        w["groups"] = self.group_bits(node.groups)
        self.group_set_id(int(w["groups"]))
        w["busy_time"] = float(node.busy_time)
        w["gpu_free"] = gfree
        w["flags"] = (NF_MAINTENANCE if node.maintenance else 0) | (NF_ACTIVE if node.active else 0) | \
                     (NF_SMT if smt else 0) | (NF_HAS_GPU if len(gpus) > 0 else 0)
        w["hp_free"] = max(-2 ** 31, min(2 ** 31 - 1, int(node.mem.free_hugepages_gb)))
        w["hp_total"] = max(-2 ** 31, min(2 ** 31 - 1, int(getattr(node.mem, "ttl_hugepages_gb", 0))))
        w["cores_per_proc"], w["numa_nodes"], w["n_gpus"] = cpp, U, len(gpus)
        return w

    def _note_share_value(self, v) -> None:
        if not _dyadic_speed(v):
            self.share_exact = False

    def admit_wire_request(self, req) -> Optional[str]:
        """ENABLE_SHARING and a request digested from a config text (nhd_amd/wire.py): its speeds are noted as _digest_fields notes a
        topology's; returns why the request cannot be answered exactly (a group with several RX / TX cores outside the exact regime),
        or None."""
        G = int(req["n_groups"])
        for g in range(G):
            self._note_share_value(req["rx"][g])
            self._note_share_value(req["tx"][g])
        fl = int(req["flags"])
        if fl & RF_NIC_SPLIT and not (fl & RF_NIC_SPLIT_DYADIC and self.share_exact):
            return ("a processing group with several RX / TX cores (ENABLE_SHARING: at most one of each per group once a speed that is not a "
                    "multiple of 2^-20 Gb/s has reached the mirror)")
        return None

    def pack_share(self, node) -> np.ndarray:
        """Node.nics[].speed_used as one nhdfit_wide_share record, NIC (numa, idx) as pack_wide orders them (nhd/Node.py:290)."""
        sh = np.zeros((), WIDE_SHARE)
        U = int(node.numa_nodes)
        cnt = [0] * WIDE_MAX_NUMA
        for nic in node.nics:
            u = nic.numa_node
            if u >= U or u < 0:
                continue
            k = cnt[u]
            sh["used"][u][k][0] = float(nic.speed_used[0])
            sh["used"][u][k][1] = float(nic.speed_used[1])
            self._note_share_value(nic.speed_used[0])
            self._note_share_value(nic.speed_used[1])
            cnt[u] = k + 1
        return sh

    def pack_nodes(self, nl: Dict[str, object]) -> NodeTable:
        t = empty_table(len(nl))
        t.names = list(nl.keys())
        self.share_exact = True                            # (every speed_used of the mirror is read again below)
        for i, node in enumerate(nl.values()):
            self.pack_node_into(node, t, i)
        return t

    # ---- K3 deltas ------------------------------------------------
    @staticmethod
    def core_masks(node, core_ids) -> Tuple[List[int], List[int]]:
        """Logical core ids -> bits of planes 0 / 1 (pack_node_into: thread-0 core u*cpp+b = bit b of t0[u]; an id in the
        sibling range is the t1 bit of the thread-0 core it belongs to, nhd/Node.py:343-350)."""
        U, cpp = int(node.sockets), int(node.cores_per_proc)
        t0, t1 = [0, 0], [0, 0]
        for c in core_ids:
            c = int(c)
            if 0 <= c < U * cpp:
                t0[c // cpp] |= 1 << (c % cpp)
            elif node.smt_enabled and U * cpp <= c < 2 * U * cpp:
                h = int(node.cores[c].sibling)
                t1[h // cpp] |= 1 << (h % cpp)
            else:
                raise UnsupportedNode(f"node {node.name}: core id {c} outside the node")
        return t0, t1

    def delta_from_topology(self, index: int, node, top, op: int) -> np.ndarray:
        """The nhdfit_delta of Node.RemoveResourcesFromTopology (DELTA_TAKE) / AddResourcesFromTopology (DELTA_GIVE) for
        a topology with physical ids (nhd/Node.py:530-636): cores, GPUs by device id, one pods_used step per
        nic_core_pairing entry whose MAC is on the node, hugepages."""
        d = np.zeros((), DELTA)
        d["node"], d["op"] = index, op
        ids = []
        gmask = 0
        pos = {g.device_id: k for k, g in enumerate(node.gpus)}
        for pg in top.proc_groups:
            ids += [c.core for c in pg.misc_cores] + [c.core for c in pg.proc_cores]
            for g in pg.group_gpus:
                k = pos.get(g.device_id)
                if k is not None:                            # "Cannot find GPU device ID": logged, skipped (Node.py:549-551)
                    gmask |= 1 << k
                ids += [c.core for c in g.cpu_cores]
        ids += [c.core for c in top.misc_cores]
        d["t0"], d["t1"] = self.core_masks(node, ids)
        d["gpus"] = gmask
        d["hugepages_gb"] = max(-2 ** 31, min(2 ** 31 - 1, int(top.hugepages_gb)))
        by_mac = {n.mac: n for n in node.nics}
        U = int(node.sockets)
        k = 0
        for p in getattr(top, "nic_core_pairing", ()):
            nic = by_mac.get(p.mac)
            if nic is None or not (0 <= nic.numa_node < U):   # not on this node / invisible to the NIC stage
                continue
            if k >= DELTA_MAX_NICS:
                raise UnsupportedNode("more than %d NIC pairings in one topology" % DELTA_MAX_NICS)
            d["nic"][k] = (int(nic.numa_node) << 4) | int(nic.idx)
            k += 1
        d["nic_n"] = k
        return d

    def delta_scalar(self, index: int, node, what: str) -> np.ndarray:
        """Deltas of the scheduler's writes to scalar node fields, from the node's state AFTER the write."""
        d = np.zeros((), DELTA)
        d["node"] = index
        if what in ("active", "maintenance"):
            d["op"] = DELTA_SET_FLAGS
            d["flags_mask"] = NF_MAINTENANCE | NF_ACTIVE
            d["flags_value"] = (NF_MAINTENANCE if node.maintenance else 0) | (NF_ACTIVE if node.active else 0)
        elif what == "groups":
            d["op"] = DELTA_SET_GROUPS
            gb = self.group_bits(node.groups)
            d["groups"], d["group_set"] = gb, self.group_set_id(gb)
        elif what == "busy_time":
            d["op"] = DELTA_SET_BUSY
            d["busy_time"] = float(node.busy_time)
        elif what == "hugepages":
            d["op"] = DELTA_SET_HUGEPAGES
            d["hugepages_gb"] = max(-2 ** 31, min(2 ** 31 - 1, int(node.mem.free_hugepages_gb)))
            d["hp_total"] = max(-2 ** 31, min(2 ** 31 - 1, int(node.mem.ttl_hugepages_gb)))
        elif what == "reset":
            d["op"] = DELTA_RESET
        else:
            raise ValueError(what)
        return d

    # ---- request side ---------------------------------------------------------------------
    def _digest_fields(self, top, pod_groups: Optional[Sequence[str]] = None, big: bool = False) -> tuple:
        """The record's 38 scalars in REQ's field order (plain Python arithmetic: one struct.pack instead of ~30 numpy field
        writes - the digest is on FindNode's per-pod path).  big=True: the scalars of a nhdfit_big_req (eight groups, BIG_REQ's
        field order) - a pod with 5..8 processing groups, answered by the general path."""
        groups = top.proc_groups
        G = len(groups)
        mt = getattr(top.map_type, "value", top.map_type)
        map_type = int(mt) if isinstance(mt, (int, np.integer)) else 0
        W = BIG_MAX_GROUPS if big else MAX_GROUPS
        if G > W:
            raise UnsupportedNode(f"pod with {G} proc groups (> {W})")
        hp = int(top.hugepages_gb)
        if hp > MAX_HUGEPAGES_GB and not big:
            raise UnsupportedNode(f"pod asks for {hp} GiB of hugepages (> {MAX_HUGEPAGES_GB}: the hugepage table of a pod tile)")
        gpus, cpu_smt, cpu_nosmt, procs, helps = [0] * W, [0] * W, [0] * W, [0] * W, [0] * W
        rxs, txs = [0.0] * W, [0.0] * W
        smt_bits = nic_use = 0
        split, split_dyadic = False, True
        for i, pg in enumerate(groups):
            n_proc = len(pg.proc_cores) + sum(len(g.cpu_cores) for g in pg.group_gpus)
            n_help = len(pg.misc_cores)
            gpus[i] = len(pg.group_gpus)
            if n_proc > 255 or n_help > 255:
                raise UnsupportedNode("a proc group asks for more than 255 cores")
            procs[i] = n_proc
            helps[i] = n_help
            p_smt, h_smt = _enum_value(pg.proc_smt), _enum_value(pg.helper_smt)
            if p_smt:
                smt_bits |= 1 << i
            if h_smt:
                smt_bits |= 1 << (W + i)
            cpu_nosmt[i] = n_proc + n_help
            cpu_smt[i] = ((n_proc + 1) // 2 if p_smt else n_proc) + ((n_help + 1) // 2 if h_smt else n_help)   # ceil(n / 2.0)
            rx = tx = 0
            n_rx = n_tx = 0
            for c in pg.proc_cores:
                d = c.nic_dir
                try:
                    d = d._value_                                   # (Enum member: the plain attribute behind .value)
                except AttributeError:
                    d = getattr(d, "value", d)
                if d == 1:
                    rx += c.nic_speed
                    n_rx += 1
                    nic_use |= 1 << i
                    if self.sharing:
                        self._note_share_value(c.nic_speed)
                elif d == 2:
                    tx += c.nic_speed
                    n_tx += 1
                    nic_use |= 1 << i
                    if self.sharing:
                        self._note_share_value(c.nic_speed)
            if n_rx > 1 or n_tx > 1:
                split = True
                split_dyadic = split_dyadic and all(_dyadic_speed(c.nic_speed) for c in pg.proc_cores
                                                    if getattr(c.nic_dir, "value", c.nic_dir) in (1, 2))
            if self.sharing and (n_rx > 1 or n_tx > 1) and not self.share_exact:
                # ENABLE_SHARING: the commit step adds every RX / TX core's speed to speed_used one after the other (nhd/Node.py:754);
                # the request record carries a group's sums - the same f64 value while a direction has one core, or while every
                # partial sum is exact (Packer.share_exact); neither holds here
                raise UnsupportedNode(f"processing group {i}: {n_rx} RX and {n_tx} TX cores (ENABLE_SHARING: at most one of each per group "
                                      "once a speed that is not a multiple of 2^-20 Gb/s has reached the mirror)")
            rxs[i] = float(rx)
            txs[i] = float(tx)
        n_misc = len(top.misc_cores)
        flags, gbits = (RF_INITIAL_FILTER, self.group_bits_known(pod_groups)) if pod_groups is not None else (0, 0)
        if split:
            flags |= RF_NIC_SPLIT | (RF_NIC_SPLIT_DYADIC if split_dyadic else 0)
        if big:
            return (G, map_type, max(-2 ** 31, min(2 ** 31 - 1, hp)), flags, int(gbits), *gpus, *cpu_smt, *cpu_nosmt,
                    (n_misc + 1) // 2 if top.misc_cores_smt else n_misc, n_misc, smt_bits, min(n_misc, 255),
                    1 if getattr(top.misc_cores_smt, "value", top.misc_cores_smt) == 1 else 0, *rxs, *txs, *procs, *helps, nic_use)
        return (G, map_type, max(-2 ** 31, min(2 ** 31 - 1, hp)), flags, int(gbits), *gpus, *cpu_smt, *cpu_nosmt,
                (n_misc + 1) // 2 if top.misc_cores_smt else n_misc,                       # Enum truthiness, quirk Q1
                n_misc, *procs, *rxs, *txs, *helps, min(n_misc, 255), smt_bits,
                1 if getattr(top.misc_cores_smt, "value", top.misc_cores_smt) == 1 else 0, nic_use)

    def digest(self, top, pod_groups: Optional[Sequence[str]] = None) -> np.ndarray:
        """CfgTopology -> nhdfit_req (nhd/CfgTopology.py:199-232 + nhd/Matcher.py:178-204).

        pod_groups given  -> the kernel applies InitialNodeFilter (NHDScheduler.py:235-247) itself;
        pod_groups None   -> the caller already filtered (`nl` of FindNode) and passes a candidate mask.
        """
        return self.digest_many([top], None if pod_groups is None else [pod_groups])[0]

    def digest_big(self, top, pod_groups: Optional[Sequence[str]] = None) -> np.ndarray:
        """CfgTopology -> nhdfit_big_req: a pod with up to BIG_MAX_GROUPS processing groups (the reference enumerates
        repeat=len(req) for any group count, nhd/Matcher.py:118,203,242), for the general path (nhdfit_big_find)."""
        out = np.zeros(1, BIG_REQ)
        try:
            _BIG_REQ_STRUCT.pack_into(memoryview(out).cast("B"), 0, *self._digest_fields(top, pod_groups, big=True))
        except struct.error as e:
            raise UnsupportedNode(str(e)) from None
        return out[0]

    def digest_many(self, tops: Sequence[object], pod_groups: Optional[Sequence[Sequence[str]]] = None,
                    unsupported: Optional[List[Tuple[int, str]]] = None) -> np.ndarray:
        """One record per pod.  A request beyond the record's limits (more than MAX_GROUPS proc groups, > 255 cores in a
        group) raises UnsupportedNode - or, with `unsupported` (a list), becomes a record that matches nothing
        (map type 0, nhd/Matcher.py:45-47) and is reported there as (index, reason)."""
        out = np.zeros(len(tops), REQ)
        raw = memoryview(out).cast("B") if len(tops) else None
        for i, top in enumerate(tops):
            try:
                _REQ_STRUCT.pack_into(raw, i * REQ.itemsize, *self._digest_fields(top, None if pod_groups is None else pod_groups[i]))
            except (UnsupportedNode, struct.error) as e:           # struct.error: a count beyond its field (65 536 misc cores ...)
                if unsupported is None or self.strict:
                    if isinstance(e, struct.error):
                        raise OverflowError(str(e)) from None
                    raise
                raw[i * REQ.itemsize:(i + 1) * REQ.itemsize] = bytes(REQ.itemsize)     # (a half-packed record must not match anything)
                unsupported.append((i, str(e)))
        return out


def expand_batch(take: int, pair: int, numa: int, cores_per_proc: int, num_cores: int, late: int = 0) -> List[int]:
    """The list one GetFreeCpuBatch call returns (nhd/Node.py:502-519) from the masks of a placement record:
    ascending physical core b of socket `numa` -> logical id numa * cores_per_proc + b, followed by its SMT
    sibling (id + num_cores, nhd/Node.py:343-350) when the pair bit is set; then the siblings the run-on walk over
    the sibling range handed out as cores of their own (`late`, ascending)."""
    out: List[int] = []
    take, pair, late = int(take), int(pair), int(late)
    base = numa * cores_per_proc
    while take:                                       # set bits, ascending
        low = take & -take
        core = base + low.bit_length() - 1
        out.append(core)
        if pair & low:
            out.append(core + num_cores)
        take ^= low
    base += num_cores
    while late:
        low = late & -late
        out.append(base + low.bit_length() - 1)
        late ^= low
    return out


# one C call turns a placement record into Python ints (a field access on a numpy record costs about as much as the whole expansion):
# 18 mask words (proc_take[4], proc_pair[4], help_take[4], help_pair[4], misc_take, misc_pair), gpu[4][8], numa[5], status, 9 more words
# (proc_late[4], help_late[4], misc_late) - the field order of PLACEMENT
_PLACEMENT_STRUCT = struct.Struct("<18Q32B5bB2x9Q")
assert _PLACEMENT_STRUCT.size == PLACEMENT.itemsize and PLACEMENT.names == (
    "proc_take", "proc_pair", "help_take", "help_pair", "misc_take", "misc_pair", "gpu", "numa", "status", "pad", "proc_late", "help_late", "misc_late")


_MAPPING_STRUCT = struct.Struct("<4b5b4b4bb2x")          # gpu[4], cpu[5], nic_numa[4], nic_idx[4], valid - the field order of MAPPING
assert _MAPPING_STRUCT.size == MAPPING.itemsize and MAPPING.names == ("gpu", "cpu", "nic_numa", "nic_idx", "valid", "pad")


def unpack_mappings(maps: np.ndarray) -> List[tuple]:
    """MAPPING records -> (gpu[4], cpu[5], nic_numa[4], nic_idx[4], valid) as tuples of Python ints per record, in one pass."""
    return [(r[0:4], r[4:9], r[9:13], r[13:17], r[17]) for r in _MAPPING_STRUCT.iter_unpack(np.ascontiguousarray(maps, dtype=MAPPING).tobytes())]


_BIG_MAPPING_STRUCT = struct.Struct("<8b9b8b8bb2x")
assert _BIG_MAPPING_STRUCT.size == BIG_MAPPING.itemsize and BIG_MAPPING.names == ("gpu", "cpu", "nic_numa", "nic_idx", "valid", "pad")


def unpack_big_mappings(maps: np.ndarray) -> List[tuple]:
    """BIG_MAPPING records (eight groups) in unpack_mappings' form."""
    return [(r[0:8], r[8:17], r[17:25], r[25:33], r[33]) for r in _BIG_MAPPING_STRUCT.iter_unpack(np.ascontiguousarray(maps, dtype=BIG_MAPPING).tobytes())]


def unpack_placements(places: np.ndarray) -> List[tuple]:
    """PLACEMENT records -> flat tuples of Python ints for expand_placement (a caller with many records converts the array once)."""
    return list(_PLACEMENT_STRUCT.iter_unpack(np.ascontiguousarray(places, dtype=PLACEMENT).tobytes()))


def expand_placement(place, n_groups: int, cores_per_proc: int, num_cores: int, gpus_per_group: Sequence[int]) -> dict:
    """nhdfit_placement -> the physical ids Node.SetPhysicalIdsFromMapping hands out, in its order:
    {'groups': [{'cores': [...], 'helpers': [...], 'gpus': [positions in Node.gpus]}], 'misc': [...]}.
    `cores` is the group's whole batch: the reference gives its first entries to the GPUs' cpu_cores (in GPU order),
    the rest to proc_cores (nhd/Node.py:729-742).  `place`: one PLACEMENT record, or its tuple from unpack_placements."""
    f = place if isinstance(place, tuple) else _PLACEMENT_STRUCT.unpack(np.asarray(place, dtype=PLACEMENT).tobytes())
    groups = []
    for g in range(n_groups):
        u = f[50 + g]
        groups.append({"cores": expand_batch(f[g], f[4 + g], u, cores_per_proc, num_cores, f[56 + g]),
                       "helpers": expand_batch(f[8 + g], f[12 + g], u, cores_per_proc, num_cores, f[60 + g]),
                       "gpus": list(f[18 + 8 * g:18 + 8 * g + gpus_per_group[g]])})
    misc = expand_batch(f[16], f[17], f[50 + MAX_GROUPS], cores_per_proc, num_cores, f[64])
    return {"groups": groups, "misc": misc}


def _mask2(words) -> int:
    return int(words[0]) | (int(words[1]) << 64)


def expand_wide_placement(place, n_groups: int, cores_per_proc: int, num_cores: int, gpus_per_group: Sequence[int]) -> dict:
    """nhdfit_wide_placement -> the ids of expand_placement (two mask words per batch)."""
    groups = []
    for g in range(n_groups):
        u = int(place["numa"][g])
        groups.append({"cores": expand_batch(_mask2(place["proc_take"][g]), _mask2(place["proc_pair"][g]), u, cores_per_proc, num_cores, _mask2(place["proc_late"][g])),
                       "helpers": expand_batch(_mask2(place["help_take"][g]), _mask2(place["help_pair"][g]), u, cores_per_proc, num_cores, _mask2(place["help_late"][g])),
                       "gpus": [int(x) for x in place["gpu"][g][:gpus_per_group[g]]]})
    misc = expand_batch(_mask2(place["misc_take"]), _mask2(place["misc_pair"]), int(place["numa"][len(place["numa"]) - 1]), cores_per_proc, num_cores,
                        _mask2(place["misc_late"]))                       # (numa[-1]: nhdfit_wide_placement has five entries, nhdfit_big_placement nine)
    return {"groups": groups, "misc": misc}


def resolve_signatures(packer: Packer, table: NodeTable):
    """Signature ids -> their interned pool tuples (for comparing tables built by different packers)."""
    caps = packer.caps

    def res(sig):
        return tuple(sorted((gl, tuple(sorted((caps[c], m) for c, m in pairs))) for gl, pairs in packer.sigs[sig]))
    return [[res(int(s)) for s in row] for row in np.concatenate([table.p3["sig_numa"], table.p3["sig_pci"]], axis=1)]
