"""Mode B (sequential commit) oracle for BASELINE-size batches.

*** TEST INFRASTRUCTURE - NOT PRODUCT CODE. ***  Only tests/, __graft_entry__.smoke() and bench.py's checker legs may
import this module.  Nothing under nhd_amd/ does, and it shares no code or data layout with the HIP path (no bitmaps,
no NIC signatures, no table rows: per-core / per-GPU / per-NIC records and explicit enumeration).

The scheduler matches pending pods one after the other and commits every winner before it matches the next
(nhd/NHDScheduler.py:425-437 over AttemptScheduling, :249-304).  `oracle.nhd_oracle.schedule_sequence` restates that loop
in pure Python on node objects (pinned to the unmodified reference by tests/test_mode_b_oracle.py) but needs
~100 us .. 1 ms per (pod, node) pair - hours at 65 536 nodes x 4 096 pods.  This module is the same loop with the two
O(N) pieces in C (oracle/nhd_oracle.c):

  * the scan for the winner: `oracle_first_feasible` (explicit enumeration per node, first feasible node in candidate order;
    a pod without GPUs first looks among the nodes with no GPU installed - SelectNode, nhd/Matcher.py:393-421);
  * the commit on the flat records: `oracle_commit` (SetBusy, SetPhysicalIdsFromMapping, ClaimPodNICResources).

The winner's *mapping* depends on CPython set order (GetNumaGroupIdx, nhd/Matcher.py:423-452), so it is asked of the
pure-Python oracle: the winner alone is materialised as a node object from the flat records and handed to
`nhd_oracle.find_node` (one node: ~1 ms).

Pinned by tests/test_seq_oracle.py: against `nhd_oracle.schedule_sequence` (decisions, mappings, physical ids, final node
state) on seeded clusters, against the reference-generated fixtures tests/golden/commit/*.json, and - in the build
container - against the unmodified reference's own loop.
"""
from __future__ import annotations

import ctypes
import os
from types import SimpleNamespace
from typing import List, Optional, Sequence

import numpy as np

from . import coracle
from . import nhd_oracle as O

_proto_done = False


def scan_threads() -> int:
    """Threads for the first-feasible scan: it hands out 256-node blocks in ascending order and stops at the first hit, so a handful is
    all it can use - a parallel region over every core of a 256-core host costs more to start than the scan itself - and never more
    than this process may run on."""
    return int(max(1, min(16, coracle.usable_cpus())))


def set_scan_threads():
    """(Re)set the OpenMP thread count for the scan.  Called at the start of every sequence, not once per process: coracle.find(threads=N)
    sets the count for the whole library, and a sequence that follows a 256-thread snapshot find in the same process would otherwise
    open a 256-thread parallel region twice per pod (found when a new test put such a find in front of the mode-B tests: 8 x slower)."""
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(scan_threads())
    except OSError:
        pass


def _lib():
    global _proto_done
    L = coracle.lib()
    if not _proto_done:
        set_scan_threads()
        L.oracle_first_feasible.restype = ctypes.c_int64
        L.oracle_first_feasible.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_int]
        L.oracle_commit.restype = ctypes.c_int
        L.oracle_commit.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
        _proto_done = True
    return L


class SeqCluster:
    """A coracle.Cluster whose flat records are committed to in place."""

    def __init__(self, cluster: coracle.Cluster, names: Optional[Sequence[str]] = None, group_names: Optional[dict] = None):
        self.c = cluster
        self.n = cluster.n
        self.nodes = np.ascontiguousarray(cluster.nodes)
        self.a = {k: np.ascontiguousarray(v) for k, v in cluster.arrays.items()}
        self.names = list(names) if names is not None else None
        self._oc = coracle._OCluster(self.nodes.ctypes.data, self.n, *[self.a[k].ctypes.data for k in
                                     ("core_used", "core_socket", "core_sibling", "gpu_used", "gpu_numa", "gpu_sw", "nic_numa",
                                      "nic_speed", "nic_pods", "nic_sw")])

    def name(self, i: int) -> str:
        return self.names[i] if self.names is not None else f"node{i:06d}"

    def node_object(self, i: int):
        """Node i as an attribute bag with the fields nhd_oracle.evaluate_node reads (nhd/Node.py:110-131)."""
        nd = self.nodes[i]
        co, nc = int(nd["core_off"]), int(nd["n_cores"])
        go, ng = int(nd["gpu_off"]), int(nd["n_gpus"])
        no, nn = int(nd["nic_off"]), int(nd["n_nics"])
        a = self.a
        U = int(nd["numa_nodes"])
        cores = [SimpleNamespace(core=k, socket=int(a["core_socket"][co + k]), sibling=int(a["core_sibling"][co + k]),
                                 used=bool(a["core_used"][co + k])) for k in range(nc)]
        gpus = [SimpleNamespace(used=bool(a["gpu_used"][go + k]), numa_node=int(a["gpu_numa"][go + k]), pciesw=int(a["gpu_sw"][go + k]))
                for k in range(ng)]
        nics, per = [], {}
        for k in range(nn):
            u = int(a["nic_numa"][no + k])
            nics.append(SimpleNamespace(numa_node=u, speed=float(a["nic_speed"][no + k]), pods_used=int(a["nic_pods"][no + k]),
                                        pciesw=int(a["nic_sw"][no + k]), idx=per.get(u, 0)))     # per-NUMA ordinal, nhd/Node.py:413-418
            per[u] = per.get(u, 0) + 1
        return SimpleNamespace(name=self.name(i), numa_nodes=U, sockets=U, cores_per_proc=int(nd["n_scan"]) // max(U, 1),
                               smt_enabled=bool(nd["smt"]), maintenance=bool(nd["maintenance"]), active=bool(nd["active"]),
                               busy_time=float(nd["busy_time"]), cores=cores, gpus=gpus, nics=nics,
                               mem=SimpleNamespace(free_hugepages_gb=int(nd["hp_free"])))

    def first_feasible(self, pod: np.ndarray, now: float, only_nogpu: bool) -> int:
        return int(_lib().oracle_first_feasible(ctypes.byref(self._oc), pod.ctypes.data_as(ctypes.c_void_p), float(now), int(only_nogpu)))

    def commit(self, i: int, pod: np.ndarray, top, mapping: dict, now: float):
        """Returns (status, ids) with ids in nhd_oracle.commit's format; status 1 = the reference raises."""
        G = int(pod["G"][0])
        mnuma = np.asarray(mapping["gpu"], np.int32)
        nnuma = np.asarray([x[0] for x in mapping["nic"]], np.int32)
        nidx = np.asarray([x[1] for x in mapping["nic"]], np.int32)
        use = np.zeros(max(G, 1), np.int32)
        for g, pg in enumerate(top.proc_groups):
            use[g] = int(any(getattr(c.nic_dir, "value", c.nic_dir) in (1, 2) for c in pg.proc_cores))
        ids = np.zeros(4096, np.int32)
        counts = np.zeros(3 * coracle.MAXG + 1, np.int32)
        smt_on = int(getattr(top.misc_cores_smt, "value", top.misc_cores_smt) == 1)          # the REAL flag, nhd/Node.py:799
        rc = _lib().oracle_commit(ctypes.byref(self._oc), int(i), pod.ctypes.data_as(ctypes.c_void_p), mnuma.ctypes.data_as(ctypes.c_void_p),
                                  int(mapping["cpu"][-1]), nnuma.ctypes.data_as(ctypes.c_void_p), nidx.ctypes.data_as(ctypes.c_void_p),
                                  use.ctypes.data_as(ctypes.c_void_p), smt_on, float(now), ids.ctypes.data_as(ctypes.c_void_p),
                                  counts.ctypes.data_as(ctypes.c_void_p))
        if rc:
            return rc, None
        out, at = {"groups": [], "misc": []}, 0
        for g in range(G):
            nc_, nh, ngp = (int(x) for x in counts[3 * g:3 * g + 3])
            cores = [int(x) for x in ids[at:at + nc_]]; at += nc_
            helpers = [int(x) for x in ids[at:at + nh]]; at += nh
            gpus = [int(x) for x in ids[at:at + ngp]]; at += ngp
            out["groups"].append({"cores": cores, "helpers": helpers, "gpus": gpus})
        out["misc"] = [int(x) for x in ids[at:at + int(counts[3 * G])]]
        return 0, out


def schedule_sequence(sc: SeqCluster, tops, pod_groups, now: float, stop_at_raise: bool = True):
    """The scheduler loop over `tops` (caller's order).  Returns (winner index or -1 per pod, mapping dict or None per pod,
    physical ids or None per pod, n_defined): pods [0, n_defined) are decided under defined reference behaviour; the first
    pod whose commit the reference would raise on ends the sequence (its unwind path is itself broken)."""
    pods = sc.c.pods_from_tops(tops, pod_groups)
    _lib()
    set_scan_threads()
    winners: List[int] = []
    maps: List[Optional[dict]] = []
    ids_all: List[Optional[dict]] = []
    for k, top in enumerate(tops):
        pod = pods[k:k + 1]
        any_gpu = any(len(pg.group_gpus) > 0 for pg in top.proc_groups)
        w = -1
        if not any_gpu:                                          # first candidate with no GPU installed, Matcher.py:405-413
            w = sc.first_feasible(pod, now, True)
        if w < 0:
            w = sc.first_feasible(pod, now, False)
        if w < 0:
            winners.append(-1); maps.append(None); ids_all.append(None)
            continue
        node = sc.node_object(w)
        res = O.find_node({node.name: node}, top, now)
        assert res[0] == node.name, ("C scan and Python evaluation disagree on feasibility", k, w)
        rc, ids = sc.commit(w, pod, top, res[1], now)
        if rc:
            if stop_at_raise:
                return winners, maps, ids_all, k
            winners.append(w); maps.append(res[1]); ids_all.append(None)
            continue
        winners.append(w); maps.append(res[1]); ids_all.append(ids)
    return winners, maps, ids_all, len(tops)
