"""Seeded synthetic clusters and pending-pod batches (SURVEY.md §8d / BASELINE.json configs).

Everything is drawn vectorised from ``numpy.random.default_rng(0x4E4844 + cfg)``
so that a 262 144-node cluster costs milliseconds and is identical on every
machine.  A cluster is held as a :class:`ClusterSpec` (plain numpy columns).  It
can be turned into

* NFD label dicts + occupancy, to be parsed by the reference's own
  ``Node.ParseLabels`` (nhd/Node.py:468) or by :mod:`workload.refmodel` when the
  reference is not installed  ->  ``spec.build_nodes(...)``; these objects feed
  ``pack.pack_nodes`` exactly like live scheduler state would;
* packed device planes directly (``workload.planes.planes_from_spec``), for cluster sizes at
  which building ~100 Python objects per node would dominate the benchmark.
  tests/test_pack.py asserts both routes give bit-identical planes.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import refmodel

SEED_BASE = 0x4E4844
GROUP_NAMES = ["default"] + [f"grp{k:02d}" for k in range(1, 16)]
SWITCH_IDS = (0x10, 0x11, 0x20, 0x21)     # GPU i / NIC-or-PF i share switch i (c4, c5)
NFD = refmodel.NFD

#            N        P      gpu_frac  nics/numa  pod_gpu_p  bw      pci_p  groups
CONFIGS = {
    1: dict(nodes=32,     pods=1,     gpu_nodes=0.0,  nics=1, gpu_p=0.0, bw=False, pci_p=0.0, ngroups=1),
    2: dict(nodes=4096,   pods=256,   gpu_nodes=0.5,  nics=1, gpu_p=0.0, bw=False, pci_p=0.0, ngroups=1),
    3: dict(nodes=16384,  pods=1024,  gpu_nodes=0.75, nics=2, gpu_p=0.7, bw=True,  pci_p=0.0, ngroups=1),
    4: dict(nodes=65536,  pods=4096,  gpu_nodes=0.75, nics=2, gpu_p=0.7, bw=True,  pci_p=0.5, ngroups=1),
    5: dict(nodes=262144, pods=16384, gpu_nodes=0.75, nics=8, gpu_p=0.7, bw=True,  pci_p=0.5, ngroups=16),
}


@dataclass
class ClusterSpec:
    cfg: int
    n: int
    nics_per_numa: int
    sriov: bool
    phys: np.ndarray          # int32   physical cores per node (2 sockets)
    smt: np.ndarray           # bool
    core_used: np.ndarray     # uint64 [n,2] bit i = physical core i of that socket occupied (reserved included)
    n_gpus: np.ndarray        # int32   0 or 4
    gpu_used: np.ndarray      # uint32  bit g = GPU g in use
    nic_used: np.ndarray      # uint32  bit (numa*K + k) = that NIC has pods_used == 1
    hp_free: np.ndarray       # int32
    maintenance: np.ndarray   # bool
    active: np.ndarray        # bool
    busy: np.ndarray          # bool
    group_bits: np.ndarray    # uint32 bit k -> GROUP_NAMES[k] (order of names = ascending k)
    name_base: int = 0        # node i is named f"node{name_base+i:06d}"
    clock_now: float = 1.0e6  # virtual monotonic clock of the snapshot
    _names: Optional[List[str]] = field(default=None, repr=False)

    # -- names / groups -------------------------------------------------
    def name(self, i: int) -> str:
        return f"node{self.name_base + i:06d}"

    def groups(self, i: int) -> List[str]:
        b = int(self.group_bits[i])
        return [GROUP_NAMES[k] for k in range(16) if b >> k & 1]

    def busy_time(self, i: int) -> float:
        # busy nodes were deployed to 5 s ago, idle ones 1000 s ago (MIN_BUSY_SECS = 30, nhd/Node.py:107)
        return self.clock_now - (5.0 if self.busy[i] else 1000.0)

    # -- NFD labels -----------------------------------------------------
    def labels(self, i: int) -> Dict[str, str]:
        c = int(self.phys[i])
        cpp = c // 2
        lab = {NFD + "nfd-extras-cpu.numSockets": "2", NFD + "nfd-extras-cpu.num_cores": str(c)}
        spans = [(2, cpp - 1), (cpp + 2, c - 1)]
        if self.smt[i]:
            lab[NFD + "cpu-hardware_multithreading"] = "true"
            spans += [(c + 2, c + cpp - 1), (c + cpp + 2, 2 * c - 1)]
        lab[NFD + "nfd-extras-cpu.isolcpus"] = "_".join(f"{a}-{b}" for a, b in spans)
        k = self.nics_per_numa
        for numa in range(2):
            if self.sriov:
                # two physical functions per NUMA node, four virtual functions each;
                # the PF itself carries a nic label and must be skipped (nhd/Node.py:380-395)
                for pf in range(2):
                    sw = SWITCH_IDS[numa * 2 + pf]
                    pfname = f"pf{numa}{pf}"
                    lab[NFD + f"nfd-extras-sriov.pf.{pfname}"] = "true"
                    lab[NFD + f"nfd-extras-nic.{pfname}.mlx.{0xA0 + numa * 2 + pf:012x}.100000Mbs.{numa}.{sw:x}.{pf}.0"] = "true"
                    for vf in range(k // 2):
                        j = pf * (k // 2) + vf
                        mac = 0x100000 + i * 64 + numa * 16 + j
                        lab[NFD + f"nfd-extras-nic.vf{numa}{j}.mlx.{mac:012x}.100000Mbs.{numa}.{sw:x}.{pf}.{vf + 1}"] = "true"
            else:
                for j in range(k):
                    sw = SWITCH_IDS[numa * 2 + (j % 2)]
                    mac = 0x100000 + i * 64 + numa * 16 + j
                    lab[NFD + f"nfd-extras-nic.eth{numa}{j}.mlx.{mac:012x}.100000Mbs.{numa}.{sw:x}.{j}.0"] = "true"
        for g in range(int(self.n_gpus[i])):
            lab[NFD + f"nfd-extras-gpu.{g}.V100.{g // 2}.{SWITCH_IDS[g]:x}"] = "true"
        lab["DATA_PLANE_VLAN"] = "100"
        lab["DATA_DEFAULT_GW"] = "10.0.0.1/32"
        lab["NHD_GROUP"] = ".".join(self.groups(i))
        if self.maintenance[i]:
            lab[refmodel.MAINT_LABEL] = "scheduled"
        return lab

    # -- object route ---------------------------------------------------
    def describe(self, i: int) -> dict:
        """JSON-able node description (see refmodel.build_node)."""
        lab = self.labels(i)
        c = int(self.phys[i])
        cpp = c // 2
        used = []
        for s in range(2):
            m = int(self.core_used[i, s])
            for b in range(2, cpp):                     # bits 0,1 are the reserved cores (already used)
                if m >> b & 1:
                    used.append(s * cpp + b)
                    if self.smt[i]:
                        used.append(s * cpp + b + c)
        k = self.nics_per_numa
        nic_bits = int(self.nic_used[i])
        # node.nics order = label order = NUMA 0 NICs then NUMA 1 NICs, idx ascending
        pods_used = [nic_bits >> (numa * k + j) & 1 for numa in range(2) for j in range(k)]
        return dict(name=self.name(i), labels=lab, hugepages=[64, int(self.hp_free[i])],
                    active=bool(self.active[i]), used_cores=sorted(used),
                    used_gpus=[g for g in range(int(self.n_gpus[i])) if int(self.gpu_used[i]) >> g & 1],
                    nic_pods_used=pods_used, busy_time=self.busy_time(i))

    def build_node(self, i: int, ref=None):
        """One node object: reference ``Node`` if `ref` (oracle.ref_loader.load()) is given, else a stand-in."""
        return refmodel.build_node(self.describe(i), ref)

    def build_nodes(self, ref=None, indices: Optional[Sequence[int]] = None) -> Dict[str, object]:
        idx = range(self.n) if indices is None else indices
        return {self.name(i): self.build_node(i, ref) for i in idx}

    def shard(self, lo: int, hi: int) -> "ClusterSpec":
        kw = {f: getattr(self, f)[lo:hi] for f in ("phys", "smt", "core_used", "n_gpus", "gpu_used", "nic_used",
                                                   "hp_free", "maintenance", "active", "busy", "group_bits")}
        return ClusterSpec(cfg=self.cfg, n=hi - lo, nics_per_numa=self.nics_per_numa, sriov=self.sriov,
                           name_base=self.name_base + lo, clock_now=self.clock_now, **kw)


def make_cluster(cfg: int, n_nodes: Optional[int] = None, seed: Optional[int] = None) -> ClusterSpec:
    """Synthetic cluster of BASELINE config `cfg` (1..5); `n_nodes` overrides the config's node count."""
    p = CONFIGS[cfg]
    n = p["nodes"] if n_nodes is None else int(n_nodes)
    rng = np.random.default_rng(SEED_BASE + cfg if seed is None else seed)
    phys = rng.choice(np.array([32, 48, 64], dtype=np.int32), size=n)
    smt = rng.random(n) < 0.75
    cpp = (phys // 2).astype(np.uint64)
    occ = rng.random((n, 2, 32)) < 0.35                      # up to 32 physical cores per socket
    weights = (np.uint64(1) << np.arange(32, dtype=np.uint64))
    used = (occ * weights).sum(axis=2).astype(np.uint64)
    valid = (np.uint64(1) << cpp) - np.uint64(1)
    core_used = ((used & valid[:, None]) | np.uint64(3)).astype(np.uint64)   # cores 0,1 of each socket reserved
    if cfg == 1:
        core_used[:] = np.uint64(3)
    has_gpu = rng.random(n) < p["gpu_nodes"]
    if cfg == 2:                                             # exactly alternating halves, deterministic
        has_gpu = (np.arange(n) % 2) == 1
    n_gpus = np.where(has_gpu, 4, 0).astype(np.int32)
    gpu_used = ((rng.random((n, 4)) < 0.3) * (1 << np.arange(4))).sum(axis=1).astype(np.uint32)
    gpu_used = np.where(has_gpu, gpu_used, 0).astype(np.uint32)
    k = p["nics"]
    nic_used = ((rng.random((n, 2 * k)) < 0.3) * (1 << np.arange(2 * k))).sum(axis=1).astype(np.uint32)
    hp_free = rng.integers(0, 65, size=n).astype(np.int32)
    maintenance = rng.random(n) < 0.02
    active = ~(rng.random(n) < 0.02)
    busy = rng.random(n) < 0.05
    if cfg == 1:
        gpu_used[:] = 0
        nic_used[:] = 0
        hp_free[:] = 64
        maintenance[:] = False
        active[:] = True
        busy[:] = False
    if p["ngroups"] > 1:
        cnt = rng.integers(1, 4, size=n)
        picks = rng.integers(0, p["ngroups"], size=(n, 3))
        bits = np.zeros(n, dtype=np.uint32)
        for j in range(3):
            bits |= np.where(j < cnt, (1 << picks[:, j]), 0).astype(np.uint32)
        group_bits = bits
    else:
        group_bits = np.ones(n, dtype=np.uint32)
    return ClusterSpec(cfg=cfg, n=n, nics_per_numa=k, sriov=(cfg == 5), phys=phys.astype(np.int32), smt=smt,
                       core_used=core_used, n_gpus=n_gpus, gpu_used=gpu_used, nic_used=nic_used, hp_free=hp_free,
                       maintenance=maintenance, active=active, busy=busy, group_bits=group_bits)


def make_pods(cfg: int, n_pods: Optional[int] = None, seed: Optional[int] = None):
    """Returns (specs, pod_groups): request descriptions for refmodel.make_topology and each pod's node-group list."""
    p = CONFIGS[cfg]
    n = p["pods"] if n_pods is None else int(n_pods)
    rng = np.random.default_rng((SEED_BASE + cfg if seed is None else seed) ^ 0x5A5A5A)
    specs, pgroups = [], []
    if cfg == 1:
        grp = dict(proc=4, rx=0, tx=0, helpers=1, proc_smt=False, helper_smt=False, gpus=[])
        for _ in range(n):
            specs.append(dict(map_type="NUMA", hugepages_gb=0, misc=2, misc_smt=False, groups=[dict(grp), dict(grp)]))
            pgroups.append(["default"])
        return specs, pgroups
    g_of = rng.choice(np.array([1, 2, 3]), size=n, p=[0.5, 0.4, 0.1])
    bw_choices = np.array([0, 10, 25, 40])
    for i in range(n):
        groups = []
        for _ in range(int(g_of[i])):
            gpus = [1] if rng.random() < p["gpu_p"] else []
            groups.append(dict(proc=int(rng.integers(2, 9)), helpers=int(rng.integers(0, 3)),
                               rx=int(rng.choice(bw_choices)) if p["bw"] else 0,
                               tx=int(rng.choice(bw_choices)) if p["bw"] else 0,
                               proc_smt=bool(rng.random() < 0.5), helper_smt=bool(rng.random() < 0.5), gpus=gpus))
        specs.append(dict(map_type="PCI" if rng.random() < p["pci_p"] else "NUMA",
                          hugepages_gb=int(rng.choice(np.array([0, 2, 4, 8]))), misc=int(rng.integers(0, 3)),
                          misc_smt=bool(rng.random() < 0.5), groups=groups))
        if p["ngroups"] > 1:
            cnt = int(rng.integers(1, 3))
            pgroups.append([GROUP_NAMES[int(x)] for x in rng.choice(p["ngroups"], size=cnt, replace=False)])
        else:
            pgroups.append(["default"])
    return specs, pgroups
