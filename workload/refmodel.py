"""Structural stand-ins for the reference's boundary types.

The drop-in boundary keeps the reference's own ``nhd.Node.Node``
and ``nhd.CfgTopology.CfgTopology`` objects unchanged: the packer
(nhd_amd/pack.py) reads them purely by attribute.
On a machine that does not have the reference installed (the GPU test box)
the synthetic generators still need *something* carrying those attributes;
this module provides minimal attribute bags with the same field names.  They
have no scheduling behaviour of their own.

Attribute surface mirrored (reference file:line):
  node   : nhd/Node.py:110-131 (Node.__init__), 23-35 (NodeCore), 37-57 (NodeNic),
           62-71 (NodeMemory), 74-84 (NodeGpu)
  request: nhd/CfgTopology.py:48-55 (Core), 70-75 (GPU), 82-98 (ProcGroup),
           126-141 (CfgTopology), enums 26-45
  labels : the NFD label grammar parsed by nhd/Node.py:312-487
"""
from __future__ import annotations

import enum
from types import SimpleNamespace
from typing import Dict, List


class SMTSetting(enum.Enum):
    SMT_DISABLED = 0
    SMT_ENABLED = 1


class NICCoreDirection(enum.Enum):
    NIC_CORE_DIRECTION_NONE = 0
    NIC_CORE_DIRECTION_RX = 1
    NIC_CORE_DIRECTION_TX = 2


class TopologyMapType(enum.Enum):
    TOPOLOGY_MAP_INVALID = 0
    TOPOLOGY_MAP_NUMA = 1
    TOPOLOGY_MAP_PCI = 2
    TOPOLOGY_MAP_NONE = 3


NFD = "feature.node.kubernetes.io/"
MIN_NIC_MBPS = 11000        # nhd/Node.py:19
# the module-level switches of nhd/Node.py:18-20 under the reference's own names: the packer reads them from the module the
# node objects come from (nhd_amd/pack.py node_module_constants), so the stand-ins carry them like the real module does
NIC_BW_AVAIL_PERCENT = 0.9
SCHEDULABLE_NIC_SPEED_THRESH_MBPS = MIN_NIC_MBPS
ENABLE_SHARING = False
MAINT_LABEL = "sigproc.viasat.io/maintenance"   # nhd/Node.py:108


class StandInNode(SimpleNamespace):
    """Attribute bag with the fields of nhd.Node.Node that the hot path reads."""


def _cpuset(text: str) -> List[int]:
    out = set()
    for part in text.split(","):
        lo, _, hi = part.partition("-")
        out.update(range(int(lo), int(hi or lo) + 1))
    return sorted(out)


def node_from_labels(name: str, labels: Dict[str, str], hugepages=(0, 0), active=True):
    """Build a stand-in node from an NFD label dict.

    Mirrors the *observable result* of Node.ParseLabels + SetHugepages
    (nhd/Node.py:468-493); tests/test_refmodel.py checks field-for-field
    equality against the real parser whenever the reference is importable.
    """
    n = StandInNode(name=name, active=active, busy_time=0.0, cores=[], gpus=[], nics=[],
                    sockets=0, numa_nodes=0, smt_enabled=False, maintenance=False,
                    cores_per_proc=0, groups=["default"], reserved_cores=[], data_vlan=0,
                    gwip="0.0.0.0/32",
                    mem=SimpleNamespace(ttl_hugepages_gb=0, free_hugepages_gb=0, res_hugepages_gb=0))
    if "NHD_GROUP" in labels:
        n.groups = labels["NHD_GROUP"].split(".")
    if MAINT_LABEL in labels:
        n.maintenance = labels[MAINT_LABEL].lower() != "not_scheduled"

    n.sockets = n.numa_nodes = int(labels[NFD + "nfd-extras-cpu.numSockets"])
    phys = int(labels[NFD + "nfd-extras-cpu.num_cores"])
    n.smt_enabled = (NFD + "cpu-hardware_multithreading") in labels
    n.cores_per_proc = phys // n.sockets
    total = phys * 2 if n.smt_enabled else phys
    for c in range(total):
        sock = int(int(c % phys) // (phys / n.sockets))
        sib = (c + phys if c < phys else c - phys) if n.smt_enabled else -1
        n.cores.append(SimpleNamespace(core=c, socket=sock, sibling=sib, used=False))
    iso_key = NFD + "nfd-extras-cpu.isolcpus"
    if iso_key in labels:
        isolated = set()
        for chunk in labels[iso_key].split("_"):
            isolated.update(_cpuset(chunk))
        for c in n.cores:
            if c.core not in isolated:
                c.used = True
                n.reserved_cores.append(c.core)

    pfs = [k.split(".")[5] for k in labels if NFD + "nfd-extras-sriov" in k]
    for key in labels:
        if NFD + "nfd-extras-nic" not in key:
            continue
        f = key.split(".")
        ifname, vendor, mac, speed = f[4], f[5], f[6], f[7]
        if ifname in pfs or "Mbs" not in speed:
            continue
        mbps = int(speed[:speed.index("Mbs")])
        if mbps < MIN_NIC_MBPS:
            continue
        n.nics.append(SimpleNamespace(
            ifname=ifname, vendor=vendor, speed=mbps / 1e3, numa_node=int(f[8]),
            pciesw=int(f[9], 16), card=int(f[10], 16), port=int(f[11]),
            speed_used=[0, 0], pods_used=0, idx=0,
            mac=":".join(a + b for a, b in zip(mac[::2], mac[1::2])).upper()))
    per_numa: Dict[int, int] = {}
    for nic in n.nics:
        nic.idx = per_numa.get(nic.numa_node, 0)
        per_numa[nic.numa_node] = nic.idx + 1

    for key in labels:
        if NFD + "nfd-extras-gpu" in key:
            f = key.split(".")
            n.gpus.append(SimpleNamespace(device_id=int(f[4]), gtype=f[5], numa_node=int(f[6]),
                                          pciesw=int(f[7], 16), used=False))
    n.data_vlan = int(labels["DATA_PLANE_VLAN"])
    n.gwip = labels["DATA_DEFAULT_GW"]
    if "RES_HUGEPAGES_GB" in labels:
        n.mem.res_hugepages_gb = int(labels["RES_HUGEPAGES_GB"])
    n.mem.ttl_hugepages_gb = hugepages[0]
    n.mem.free_hugepages_gb = hugepages[1] - n.mem.res_hugepages_gb
    return n


def build_node(desc: dict, ref=None):
    """Materialise a node *description* (the JSON-able form used by tests/golden and synth):

        {"name", "labels", "hugepages": [alloc, free], "active", "used_cores": [logical ids],
         "used_gpus": [list index], "nic_pods_used": [per nic, node.nics order], "busy_time"}

    through the reference's own label parser when `ref` (oracle.ref_loader.load()) is given,
    else through :func:`node_from_labels`.  Occupancy is applied the way the scheduler leaves it
    behind (core.used / gpu.used / nic.pods_used / busy_time)."""
    if ref is not None:
        node = ref.Node(desc["name"], bool(desc.get("active", True)))
        if not node.ParseLabels(desc["labels"]):
            raise ValueError("label parser rejected node %s" % desc["name"])
        node.SetHugepages(*desc["hugepages"])
    else:
        node = node_from_labels(desc["name"], desc["labels"], tuple(desc["hugepages"]), bool(desc.get("active", True)))
    for c in desc.get("used_cores", ()):
        node.cores[c].used = True
    for g in desc.get("used_gpus", ()):
        node.gpus[g].used = True
    for nic, cnt in zip(node.nics, desc.get("nic_pods_used", ())):
        nic.pods_used = int(cnt)
    for nic, used in zip(node.nics, desc.get("nic_speed_used", ())):       # (rx, tx) Gb/s already on the NIC (nhd/Node.py:46)
        nic.speed_used = [used[0], used[1]]
    node.busy_time = float(desc.get("busy_time", 0.0))
    return node


# ----------------------------------------------------------------------------
# request side
# ----------------------------------------------------------------------------
class StandInTopology(SimpleNamespace):
    """Attribute bag with the fields of nhd.CfgTopology.CfgTopology the hot path reads."""


def make_topology(spec: dict, types=None):
    """Build a request object from a plain description.

    spec = {"map_type": "NUMA"|"PCI"|"NONE", "hugepages_gb": int, "misc": int,
            "misc_smt": bool, "groups": [ {"proc": int (>=2, first two are the RX/TX pair),
            "rx": float, "tx": float, "more_nic_pairs": [(rx, tx), ...] (optional), "helpers": int, "proc_smt": bool, "helper_smt": bool,
            "gpus": [n_cpu_cores_of_gpu0, ...]}, ... ]}

    `types` is the reference namespace from oracle.ref_loader.load() (then real
    nhd.CfgTopology objects are produced) or None (stand-ins are produced).
    """
    if types is not None:
        return _make_reference_topology(spec, types)
    mt = {"NUMA": TopologyMapType.TOPOLOGY_MAP_NUMA, "PCI": TopologyMapType.TOPOLOGY_MAP_PCI,
          "NONE": TopologyMapType.TOPOLOGY_MAP_NONE}.get(spec["map_type"], TopologyMapType.TOPOLOGY_MAP_INVALID)
    top = StandInTopology(proc_groups=[], misc_cores=[], nic_core_pairing=[], map_type=mt,
                          hugepages_gb=spec.get("hugepages_gb", 0),
                          misc_cores_smt=SMTSetting.SMT_ENABLED if spec.get("misc_smt") else SMTSetting.SMT_DISABLED,
                          ctrl_vlan=SimpleNamespace(name="ctrl", vlan=0), data_default_gw="")

    def core(nm, speed=0, d=NICCoreDirection.NIC_CORE_DIRECTION_NONE):
        return SimpleNamespace(name=nm, nic_speed=speed, nic_dir=d, numa=2, core=-1)

    for gi, g in enumerate(spec["groups"]):
        pg = SimpleNamespace(proc_cores=[], misc_cores=[], group_gpus=[], vlan=SimpleNamespace(name="d", vlan=0),
                             proc_smt=SMTSetting.SMT_ENABLED if g.get("proc_smt") else SMTSetting.SMT_DISABLED,
                             helper_smt=SMTSetting.SMT_ENABLED if g.get("helper_smt") else SMTSetting.SMT_DISABLED)
        nproc = g["proc"]
        if nproc >= 2:
            rx = core(f"g{gi}rx", g.get("rx", 0), NICCoreDirection.NIC_CORE_DIRECTION_RX)
            tx = core(f"g{gi}tx", g.get("tx", 0), NICCoreDirection.NIC_CORE_DIRECTION_TX)
            pg.proc_cores += [rx, tx]
            top.nic_core_pairing.append(SimpleNamespace(rx_core=rx, tx_core=tx, mac="", rx_ring_size=4096))
        more = g.get("more_nic_pairs", []) if nproc >= 2 else []      # further (rx speed, tx speed) pairs, out of the same `proc` cores
        for k, (rs, ts) in enumerate(more):
            rx = core(f"g{gi}rx{k + 1}", rs, NICCoreDirection.NIC_CORE_DIRECTION_RX)
            tx = core(f"g{gi}tx{k + 1}", ts, NICCoreDirection.NIC_CORE_DIRECTION_TX)
            pg.proc_cores += [rx, tx]
            top.nic_core_pairing.append(SimpleNamespace(rx_core=rx, tx_core=tx, mac="", rx_ring_size=4096))
        for k in range(max(0, nproc - 2 - 2 * len(more)) if nproc >= 2 else nproc):
            pg.proc_cores.append(core(f"g{gi}p{k}"))
        for k in range(g.get("helpers", 0)):
            pg.misc_cores.append(core(f"g{gi}h{k}"))
        for k, ncpu in enumerate(g.get("gpus", [])):
            pg.group_gpus.append(SimpleNamespace(cpu_cores=[core(f"g{gi}gpu{k}c{j}") for j in range(ncpu)],
                                                 dev_id_names=[], gtype=None, device_id=-1))
        top.proc_groups.append(pg)
    for k in range(spec.get("misc", 0)):
        top.misc_cores.append(core(f"m{k}"))
    return top


def _make_reference_topology(spec, R):
    top = R.CfgTopology()
    if spec["map_type"] in ("NUMA", "PCI"):
        top.SetTopMapType(spec["map_type"])
    elif spec["map_type"] == "NONE":
        top.map_type = R.TopologyMapType.TOPOLOGY_MAP_NONE
    top.hugepages_gb = spec.get("hugepages_gb", 0)
    top.SetCtrlVlan(R.VLANInfo("ctrl", 0))
    top.SetMiscCoreSmt(R.SMTSetting.SMT_ENABLED if spec.get("misc_smt") else R.SMTSetting.SMT_DISABLED)
    none = R.NICCoreDirection.NIC_CORE_DIRECTION_NONE
    grp = R.NUMASetting.LOGICAL_NUMA_GROUP
    for gi, g in enumerate(spec["groups"]):
        pg = R.ProcGroup()
        pg.SetDataVlan(R.VLANInfo("d", 0))
        pg.SetProcSmt(R.SMTSetting.SMT_ENABLED if g.get("proc_smt") else R.SMTSetting.SMT_DISABLED)
        pg.SetHelperSmt(R.SMTSetting.SMT_ENABLED if g.get("helper_smt") else R.SMTSetting.SMT_DISABLED)
        nproc = g["proc"]
        if nproc >= 2:
            rx = R.Core(f"g{gi}rx", g.get("rx", 0), R.NICCoreDirection.NIC_CORE_DIRECTION_RX, grp, -1)
            tx = R.Core(f"g{gi}tx", g.get("tx", 0), R.NICCoreDirection.NIC_CORE_DIRECTION_TX, grp, -1)
            pg.AddGroupCore(rx)
            pg.AddGroupCore(tx)
            top.AddNicPairing(rx, tx)
        more = g.get("more_nic_pairs", []) if nproc >= 2 else []
        for k, (rs, ts) in enumerate(more):
            rx = R.Core(f"g{gi}rx{k + 1}", rs, R.NICCoreDirection.NIC_CORE_DIRECTION_RX, grp, -1)
            tx = R.Core(f"g{gi}tx{k + 1}", ts, R.NICCoreDirection.NIC_CORE_DIRECTION_TX, grp, -1)
            pg.AddGroupCore(rx)
            pg.AddGroupCore(tx)
            top.AddNicPairing(rx, tx)
        for k in range(max(0, nproc - 2 - 2 * len(more)) if nproc >= 2 else nproc):
            pg.AddGroupCore(R.Core(f"g{gi}p{k}", 0, none, grp, -1))
        for k in range(g.get("helpers", 0)):
            pg.AddMiscCore(R.Core(f"g{gi}h{k}", 0, none, grp, -1))
        for k, ncpu in enumerate(g.get("gpus", [])):
            cores = [R.Core(f"g{gi}gpu{k}c{j}", 0, none, grp, -1) for j in range(ncpu)]
            pg.AddGroupGPU(R.GPU(cores, [], R.GpuType.GPU_TYPE_ALL, -1))
        top.AddProcGroup(pg)
    for k in range(spec.get("misc", 0)):
        top.AddMiscCore(R.Core(f"m{k}", 0, none, grp, -1))
    return top
