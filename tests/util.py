"""Shared helpers for the test-suite: heterogeneous random clusters / pods (wider than the
BASELINE configs: 1- and 2-socket nodes, mixed NIC speeds, slow NICs, odd switch layouts)."""
import numpy as np

from workload import refmodel
from workload.refmodel import NFD

CLOCK = 1.0e6


def random_labels(rng, sockets=None):
    sockets = int(rng.choice([1, 2], p=[0.15, 0.85])) if sockets is None else sockets
    phys = int(rng.choice([8, 12, 16, 24])) * sockets
    smt = rng.random() < 0.6
    lab = {NFD + "nfd-extras-cpu.numSockets": str(sockets), NFD + "nfd-extras-cpu.num_cores": str(phys)}
    if smt:
        lab[NFD + "cpu-hardware_multithreading"] = "true"
    if rng.random() < 0.7:
        cpp = phys // sockets
        spans = [(s * cpp + 1, (s + 1) * cpp - 1) for s in range(sockets)]
        if smt:
            spans += [(phys + s * cpp + 1, phys + (s + 1) * cpp - 1) for s in range(sockets)]
        lab[NFD + "nfd-extras-cpu.isolcpus"] = "_".join(f"{a}-{b}" for a, b in spans)
    switches = [0x10, 0x11, 0x20, 0x21, 0x30]
    nnic = int(rng.integers(0, 7))
    for j in range(nnic):
        numa = int(rng.integers(0, sockets))
        speed = int(rng.choice([10000, 25000, 40000, 100000], p=[0.1, 0.3, 0.2, 0.4]))
        sw = switches[numa * 2 + int(rng.integers(0, 2))] if rng.random() < 0.9 else 0x30 + numa
        lab[NFD + f"nfd-extras-nic.eth{j}.mlx.{0xABC000 + j:012x}.{speed}Mbs.{numa}.{sw:x}.{j}.0"] = "true"
    ngpu = int(rng.choice([0, 1, 2, 4], p=[0.35, 0.15, 0.25, 0.25]))
    for g in range(ngpu):
        numa = int(rng.integers(0, sockets))
        sw = switches[numa * 2 + int(rng.integers(0, 2))]
        lab[NFD + f"nfd-extras-gpu.{g}.V100.{numa}.{sw:x}"] = "true"
    lab["DATA_PLANE_VLAN"] = "7"
    lab["DATA_DEFAULT_GW"] = "10.1.0.1/32"
    if rng.random() < 0.5:
        lab["NHD_GROUP"] = ".".join(rng.choice(["default", "alpha", "beta"], size=int(rng.integers(1, 3)), replace=False))
    if rng.random() < 0.05:
        lab[refmodel.MAINT_LABEL] = "scheduled"
    return lab


def random_node_desc(rng, name, occupancy=0.3):
    """Node description (refmodel.build_node format) with random occupancy, incl. half-used SMT pairs."""
    lab = random_labels(rng)
    sockets = int(lab[NFD + "nfd-extras-cpu.numSockets"])
    phys = int(lab[NFD + "nfd-extras-cpu.num_cores"])
    smt = (NFD + "cpu-hardware_multithreading") in lab
    used = []
    for c in range(phys):
        r = rng.random()
        if r < occupancy:
            used.append(c)
            if smt and rng.random() < 0.8:
                used.append(c + phys)
        elif smt and r < occupancy + 0.05:
            used.append(c + phys)
    ngpu = sum(1 for k in lab if "nfd-extras-gpu" in k)
    nnic = sum(1 for k in lab if "nfd-extras-nic" in k and "10000Mbs" not in k.replace("100000Mbs", ""))
    return dict(name=name, labels=lab, hugepages=[16, int(rng.integers(0, 17))], active=bool(rng.random() > 0.05),
                used_cores=sorted(used), used_gpus=[g for g in range(ngpu) if rng.random() < 0.3],
                nic_pods_used=[int(rng.random() < 0.25) for _ in range(nnic)],
                busy_time=CLOCK - (3.0 if rng.random() < 0.1 else 500.0))


def random_cluster_desc(seed, n, occupancy=0.3):
    rng = np.random.default_rng(seed)
    return [random_node_desc(rng, f"n{i:04d}", occupancy) for i in range(n)]


def build_cluster(descs, ref=None):
    return {d["name"]: refmodel.build_node(d, ref) for d in descs}


def random_cluster(seed, n, ref=None, occupancy=0.3):
    return build_cluster(random_cluster_desc(seed, n, occupancy), ref)


def random_pod_spec(rng, max_groups=3):
    groups = []
    for _ in range(int(rng.integers(1, max_groups + 1))):
        ng = int(rng.choice([0, 1, 2], p=[0.5, 0.4, 0.1]))
        groups.append(dict(proc=int(rng.integers(2, 7)), helpers=int(rng.integers(0, 3)),
                           rx=float(rng.choice([0, 0, 5, 10, 22.5, 25, 40, 90, 0.1])),
                           tx=float(rng.choice([0, 0, 5, 10, 12.25, 45, 90])),
                           proc_smt=bool(rng.random() < 0.5), helper_smt=bool(rng.random() < 0.5),
                           gpus=[int(rng.integers(0, 3)) for _ in range(ng)]))
    return dict(map_type=str(rng.choice(["NUMA", "PCI", "NONE"], p=[0.55, 0.4, 0.05])),
                hugepages_gb=int(rng.choice([0, 1, 4, 8])), misc=int(rng.integers(0, 4)),
                misc_smt=bool(rng.random() < 0.5), groups=groups)


def random_wide_labels(rng):
    """Labels of a node beyond the fast layout (or at its edge): 1-4 sockets, 6-128 physical cores per socket, NICs and GPUs
    on any NUMA node, now and then a PCIe switch with NICs on two NUMA nodes."""
    sockets = int(rng.choice([1, 2, 3, 4], p=[0.1, 0.3, 0.3, 0.3]))
    cpp = int(rng.choice([6, 10, 16, 24, 72, 96, 128], p=[0.15, 0.15, 0.15, 0.15, 0.15, 0.15, 0.1]))
    phys = cpp * sockets
    smt = rng.random() < 0.6
    lab = {NFD + "nfd-extras-cpu.numSockets": str(sockets), NFD + "nfd-extras-cpu.num_cores": str(phys)}
    if smt:
        lab[NFD + "cpu-hardware_multithreading"] = "true"
    if rng.random() < 0.6:
        spans = [(s * cpp + 1, (s + 1) * cpp - 1) for s in range(sockets)]
        if smt:
            spans += [(phys + s * cpp + 1, phys + (s + 1) * cpp - 1) for s in range(sockets)]
        lab[NFD + "nfd-extras-cpu.isolcpus"] = "_".join(f"{a}-{b}" for a, b in spans)
    nnic = int(rng.integers(0, 9))
    for j in range(nnic):
        numa = int(rng.integers(0, sockets))
        speed = int(rng.choice([10000, 25000, 40000, 100000], p=[0.1, 0.3, 0.2, 0.4]))
        sw = 0x10 * (numa + 1) + int(rng.integers(0, 2)) if rng.random() < 0.85 else 0x70     # 0x70: one switch seen from every NUMA node
        lab[NFD + f"nfd-extras-nic.eth{j}.mlx.{0xABD000 + j:012x}.{speed}Mbs.{numa}.{sw:x}.{j}.0"] = "true"
    ngpu = int(rng.choice([0, 1, 2, 4, 6], p=[0.3, 0.15, 0.2, 0.2, 0.15]))
    for g in range(ngpu):
        numa = int(rng.integers(0, sockets))
        sw = 0x10 * (numa + 1) + int(rng.integers(0, 2)) if rng.random() < 0.9 else 0x70
        lab[NFD + f"nfd-extras-gpu.{g}.V100.{numa}.{sw:x}"] = "true"
    lab["DATA_PLANE_VLAN"] = "7"
    lab["DATA_DEFAULT_GW"] = "10.1.0.1/32"
    if rng.random() < 0.5:
        lab["NHD_GROUP"] = ".".join(rng.choice(["default", "alpha", "beta"], size=int(rng.integers(1, 3)), replace=False))
    if rng.random() < 0.05:
        lab[refmodel.MAINT_LABEL] = "scheduled"
    return lab


def random_wide_node_desc(rng, name, occupancy=0.3):
    lab = random_wide_labels(rng)
    phys = int(lab[NFD + "nfd-extras-cpu.num_cores"])
    smt = (NFD + "cpu-hardware_multithreading") in lab
    used = []
    for c in range(phys):
        r = rng.random()
        if r < occupancy:
            used.append(c)
            if smt and rng.random() < 0.8:
                used.append(c + phys)
        elif smt and r < occupancy + 0.05:
            used.append(c + phys)
    ngpu = sum(1 for k in lab if "nfd-extras-gpu" in k)
    nnic = sum(1 for k in lab if "nfd-extras-nic" in k and "10000Mbs" not in k.replace("100000Mbs", ""))
    return dict(name=name, labels=lab, hugepages=[16, int(rng.integers(0, 17))], active=bool(rng.random() > 0.05),
                used_cores=sorted(used), used_gpus=[g for g in range(ngpu) if rng.random() < 0.3],
                nic_pods_used=[int(rng.random() < 0.25) for _ in range(nnic)],
                busy_time=CLOCK - (3.0 if rng.random() < 0.1 else 500.0))


def mixed_cluster_desc(seed, n, wide_share=0.4, occupancy=0.3):
    """Ordinary nodes (random_node_desc) with nodes beyond the fast layout in between."""
    rng = np.random.default_rng(seed)
    return [random_wide_node_desc(rng, f"w{i:04d}", occupancy) if rng.random() < wide_share else random_node_desc(rng, f"n{i:04d}", occupancy)
            for i in range(n)]


def mixed_cluster(seed, n, ref=None, wide_share=0.4, occupancy=0.3):
    return build_cluster(mixed_cluster_desc(seed, n, wide_share, occupancy), ref)
