#!/usr/bin/env python3
"""Generate tests/golden/big/*.json by running the UNMODIFIED reference (imported from /root/reference).
TEST INFRASTRUCTURE; run in the build container only:

    python oracle/gen_golden_big.py

Pods with MORE processing groups than the table-driven pass holds (5..8; the reference enumerates
itertools.product(range(numa_nodes), repeat=len(req)) for any group count, nhd/Matcher.py:118,203,242), mixed with ordinary
pods, on clusters of ordinary nodes, of ordinary and wide nodes (tests/util.mixed_cluster_desc) and of nodes with several
interchangeable NICs per NUMA node (SR-IOV VFs of one PF: what the general path's NIC search prunes by symmetry), under
both batch semantics - the fields are those of oracle/gen_golden_wide.py:
  snapshot[i]  = Matcher().FindNode over the whole cluster for pod i, every pod against the same state (mode A)
  feasible[i]  = '0'/'1' per node: the reference places pod i on that node when it is the only candidate
  sequence[i]  = [node, mapping, ids] or [None]: the scheduler's loop (nhd/NHDScheduler.py:274-304)
  final[name]  = the node afterwards
NIC counts per node are kept small (the reference makes (sum of NICs)^G deepcopies per pod and node, nhd/Matcher.py:254).
"""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from workload import refmodel                # noqa: E402
from workload.refmodel import NFD            # noqa: E402
from oracle import ref_loader                # noqa: E402
from oracle.gen_golden_wide import jsonable, ids_of, node_state   # noqa: E402
from tests import util                       # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "big")


def cap_nics(desc, most):
    """Keep the first `most` NIC labels of a node description (and as many pods_used entries)."""
    lab, kept = {}, 0
    for k, v in desc["labels"].items():
        if "nfd-extras-nic" in k:
            if kept >= most:
                continue
            kept += 1
        lab[k] = v
    desc["labels"] = lab
    nnic = sum(1 for k in lab if "nfd-extras-nic" in k and "10000Mbs" not in k.replace("100000Mbs", ""))
    desc["nic_pods_used"] = desc["nic_pods_used"][:nnic]
    return desc


def big_pod_spec(rng, lo=5, hi=8, small_share=0.3):
    """A pod with lo..hi processing groups (or, now and then, an ordinary one): few cores per group so that nodes take it."""
    if rng.random() < small_share:
        s = util.random_pod_spec(rng, max_groups=4)
    else:
        groups = []
        G = int(rng.choice(np.arange(lo, hi + 1), p=np.array([0.4, 0.3, 0.2, 0.1][:hi - lo + 1]) / sum([0.4, 0.3, 0.2, 0.1][:hi - lo + 1])))
        for _ in range(G):
            ng = int(rng.choice([0, 1], p=[0.8, 0.2]))
            groups.append(dict(proc=int(rng.integers(2, 4)), helpers=int(rng.integers(0, 2)),
                               rx=float(rng.choice([0, 0, 5, 10, 22.5, 25, 40, 0.1])), tx=float(rng.choice([0, 0, 5, 10, 12.25, 45])),
                               proc_smt=bool(rng.random() < 0.5), helper_smt=bool(rng.random() < 0.5),
                               gpus=[int(rng.integers(0, 2)) for _ in range(ng)]))
        s = dict(map_type=str(rng.choice(["NUMA", "PCI"], p=[0.6, 0.4])), hugepages_gb=int(rng.choice([0, 1, 4])),
                 misc=int(rng.integers(0, 3)), misc_smt=True, groups=groups)
    s["misc_smt"] = True                                  # (keeps the commit away from quirk Q1's raise; tests/golden/commit covers Q1)
    if s["map_type"] == "NONE":
        s["map_type"] = "NUMA"
    return s


def vf_node_desc(rng, name, vfs):
    """Two sockets, 24 physical cores each, `vfs` interchangeable 100 GbE NICs per NUMA node behind one switch per NUMA node
    (+ one GPU per switch for the PCI pods), one NIC now and then already carrying a pod."""
    lab = {NFD + "nfd-extras-cpu.numSockets": "2", NFD + "nfd-extras-cpu.num_cores": "48", NFD + "cpu-hardware_multithreading": "true"}
    j = 0
    for numa in range(2):
        for _ in range(vfs):
            lab[NFD + f"nfd-extras-nic.vf{j}.mlx.{0xABE000 + j:012x}.100000Mbs.{numa}.{0x10 * (numa + 1):x}.{j}.0"] = "true"
            j += 1
    for g in range(4):
        lab[NFD + f"nfd-extras-gpu.{g}.V100.{g // 2}.{0x10 * (g // 2 + 1):x}"] = "true"
    lab["DATA_PLANE_VLAN"] = "7"
    lab["DATA_DEFAULT_GW"] = "10.1.0.1/32"
    return dict(name=name, labels=lab, hugepages=[16, 16], active=True, used_cores=[c for c in range(48) if rng.random() < 0.1],
                used_gpus=[g for g in range(4) if rng.random() < 0.25], nic_pods_used=[int(rng.random() < 0.2) for _ in range(2 * vfs)],
                busy_time=util.CLOCK - 500.0)


def multi_socket_desc(rng, name, sockets, cpp, nics_on, gpus_on=()):
    """`sockets` sockets of `cpp` physical cores (SMT), one 100 GbE NIC on each NUMA node of `nics_on` (switch 0x10 * (numa + 1)),
    one GPU on each NUMA node of `gpus_on` behind that NUMA node's switch."""
    phys = sockets * cpp
    lab = {NFD + "nfd-extras-cpu.numSockets": str(sockets), NFD + "nfd-extras-cpu.num_cores": str(phys), NFD + "cpu-hardware_multithreading": "true"}
    for j, numa in enumerate(nics_on):
        lab[NFD + f"nfd-extras-nic.eth{j}.mlx.{0xABF000 + j:012x}.100000Mbs.{numa}.{0x10 * (numa + 1):x}.{j}.0"] = "true"
    for g, numa in enumerate(gpus_on):
        lab[NFD + f"nfd-extras-gpu.{g}.V100.{numa}.{0x10 * (numa + 1):x}"] = "true"
    lab["DATA_PLANE_VLAN"] = "7"
    lab["DATA_DEFAULT_GW"] = "10.1.0.1/32"
    return dict(name=name, labels=lab, hugepages=[16, 16], active=True, used_cores=[c for c in range(phys) if rng.random() < 0.1],
                used_gpus=[], nic_pods_used=[0] * len(nics_on), busy_time=util.CLOCK - 500.0)


def run_case(ref, descs, specs, fname, extra=None):
    t_start = time.time()
    clock = ref_loader.VirtualClock(util.CLOCK).install()
    nl = util.build_cluster(descs, ref)
    snapshot, feas = [], []
    for k, s in enumerate(specs):
        top = refmodel.make_topology(s, ref)
        snapshot.append(jsonable(ref_loader.find_node(nl, top)))
        feas.append("".join("1" if ref_loader.find_node({name: node}, top)[0] is not None else "0" for name, node in nl.items()))
        print(f"  {fname}: pod {k} ({len(s['groups'])} groups) snapshot done, {time.time() - t_start:.0f} s", flush=True)
    sequence = []
    for s in specs:
        top = refmodel.make_topology(s, ref)
        res = ref_loader.find_node(nl, top)
        if res[0] is None:
            sequence.append([None])
            continue
        n = nl[res[0]]
        n.SetBusy()
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                nic_list = n.SetPhysicalIdsFromMapping(res[1], top)
        except IndexError:
            break
        n.ClaimPodNICResources(list({x[0] for x in nic_list}))
        sequence.append(jsonable(res) + [ids_of(top, n)])
    if len(sequence) < len(specs):                            # final state = the defined prefix replayed on fresh objects
        keep = len(sequence)
        nl = util.build_cluster(descs, ref)
        for s, want in zip(specs[:keep], sequence):
            top = refmodel.make_topology(s, ref)
            res = ref_loader.find_node(nl, top)
            assert jsonable(res) == want[:2]
            if res[0] is None:
                continue
            n = nl[res[0]]
            n.SetBusy()
            with contextlib.redirect_stdout(io.StringIO()):
                nic_list = n.SetPhysicalIdsFromMapping(res[1], top)
            n.ClaimPodNICResources(list({x[0] for x in nic_list}))
    case = {"clock": clock.t, "nodes": descs, "pods": specs, "snapshot": snapshot, "feasible": feas, "sequence": sequence,
            "final": {name: node_state(n) for name, n in nl.items()}}
    case.update(extra or {})
    with open(os.path.join(OUT, fname + ".json"), "w") as f:
        json.dump(case, f, separators=(",", ":"))
    big = [len(s["groups"]) > 4 for s in specs]
    print(fname, len(descs), "nodes;", sum(big), "of", len(specs), "pods with > 4 groups;",
          sum(1 for r, b in zip(snapshot, big) if b and r[0] is not None), "of them placed in the snapshot,",
          sum(1 for r, b in zip(sequence, big) if b and r[0] is not None), "in sequence (", len(sequence), "defined );",
          f"{time.time() - t_start:.0f} s", flush=True)


def main():
    only = sys.argv[1:]
    ref = ref_loader.load()
    os.makedirs(OUT, exist_ok=True)
    if not only or "big_plain" in only:                       # ordinary nodes only
        rng = np.random.default_rng(82001)
        descs = [cap_nics(d, 4) for d in util.random_cluster_desc(82001, 26, occupancy=0.1)]
        run_case(ref, descs, [big_pod_spec(rng) for _ in range(36)], "big_plain")
    if not only or "big_mixed" in only:                       # ordinary and wide nodes
        rng = np.random.default_rng(82002)
        descs = [cap_nics(d, 4) for d in util.mixed_cluster_desc(82002, 22, wide_share=0.4, occupancy=0.12)]
        wide = [d["name"] for d in descs if d["name"].startswith("w")]
        run_case(ref, descs, [big_pod_spec(rng, 5, 7) for _ in range(30)], "big_mixed", {"drawn_wide": wide})
    if not only or "big_vf" in only:                          # interchangeable NICs: the symmetry pruning of the NIC search
        rng = np.random.default_rng(82003)
        descs = [vf_node_desc(rng, f"v{i:04d}", 3) for i in range(4)]
        specs = [big_pod_spec(rng, 5, 6, small_share=0.0) for _ in range(8)]
        for s in specs:                                       # every group brings traffic: NICs get shared until their capacity is gone
            for g in s["groups"]:
                g["rx"] = float(rng.choice([10, 25, 40, 45]))
                g["tx"] = float(rng.choice([5, 10, 45]))
        run_case(ref, descs, specs, "big_vf")


    if not only or "big_quad" in only:                        # three and four sockets with seven and eight groups: 3^8 .. 4^9 assignment tuples
        rng = np.random.default_rng(82004)
        descs = [multi_socket_desc(rng, "w0000", 4, 12, (0, 1, 2, 3), (0, 2)), util.random_node_desc(rng, "n0001", 0.1),
                 multi_socket_desc(rng, "w0002", 3, 20, (0, 1, 2)), multi_socket_desc(rng, "w0003", 4, 16, (0, 0, 2, 3), (1,)),
                 cap_nics(util.random_node_desc(rng, "n0004", 0.1), 3), multi_socket_desc(rng, "w0005", 4, 10, (1, 3))]
        specs = []
        for k in range(8):                                    # (seven groups at most here: at 4^9 tuples the reference's own list scans,
            s = big_pod_spec(rng, 7 if k < 3 else 5, 7 if k < 3 else 6, small_share=0.0)   # nhd/Matcher.py:371-373, take hours per node)
            if len(s["groups"]) >= 7:
                s["map_type"] = "NUMA"                        # (the reference's PCI pruning is quadratic in the NIC combinations, nhd/Matcher.py:329-335)
            for g in s["groups"]:
                g["proc"], g["helpers"] = 2, 0
            specs.append(s)
        run_case(ref, descs, specs, "big_quad", {"drawn_wide": ["w0000", "w0002", "w0003", "w0005"]})


    if not only or "big_tri" in only:                         # three sockets, eight groups: 3^9 tuples
        rng = np.random.default_rng(82005)
        descs = [multi_socket_desc(rng, "w0000", 3, 24, (0, 1, 2), (0,)), multi_socket_desc(rng, "w0001", 3, 16, (0, 2, 2)),
                 cap_nics(util.random_node_desc(rng, "n0002", 0.1), 2), multi_socket_desc(rng, "w0003", 3, 32, (1, 2))]
        specs = []
        for k in range(6):
            s = big_pod_spec(rng, 8, 8, small_share=0.0)
            s["map_type"] = "NUMA" if k % 3 else "PCI"
            for g in s["groups"]:
                g["proc"], g["helpers"] = 2, 0
            specs.append(s)
        run_case(ref, descs, specs, "big_tri", {"drawn_wide": ["w0000", "w0001", "w0003"]})


if __name__ == "__main__":
    main()
