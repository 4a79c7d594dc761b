#!/usr/bin/env python3
"""Generate tests/golden/*.json by running the UNMODIFIED reference (imported from
/root/reference) on seeded inputs.  TEST INFRASTRUCTURE; run in the build container only:

    python oracle/gen_golden.py

Each fixture is {"clock", "nodes": [node descriptions, refmodel.build_node format],
"pods": [{"spec", "groups"}], "expected": [FindNode result per pod], "feasible": [per pod a
'0'/'1' string over nodes]}.  `expected[i]` is reference ``Matcher().FindNode`` applied to
``InitialNodeFilter(nodes, pods[i].groups)`` (nhd/NHDScheduler.py:274-277) with every pod matched
against the same snapshot (no commit);  `feasible[i][j]` is whether the reference places pod i on
node j when that node is the only candidate (FindNode on a one-element dict).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from workload import refmodel, synth# noqa: E402
from workload.refmodel import NFD             # noqa: E402
from oracle import ref_loader                # noqa: E402
from oracle import nhd_oracle                # noqa: E402
from tests import util                       # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def jsonable(res):
    if res[0] is None:
        return [None]
    m = res[1]
    return [res[0], {"gpu": list(m["gpu"]), "cpu": list(m["cpu"]), "nic": [list(x) for x in m["nic"]]}]


def run_case(ref, clock, descs, pods):
    vc = ref_loader.VirtualClock(clock).install()
    nl = util.build_cluster(descs, ref)
    expected, feas = [], []
    for pod in pods:
        top = refmodel.make_topology(pod["spec"], ref)
        sub = nhd_oracle.initial_node_filter(nl, pod["groups"])
        expected.append(jsonable(ref_loader.find_node(sub, top)))
        row = ""
        for name, node in nl.items():
            one = {name: node} if name in sub else {}
            row += "1" if ref_loader.find_node(one, top)[0] is not None else "0"
        feas.append(row)
    del vc
    return {"clock": clock, "nodes": descs, "pods": pods, "expected": expected, "feasible": feas}


def node_desc(name, sockets=2, phys=32, smt=True, nics=((0, 100000, 0x10), (1, 100000, 0x20)),
              gpus=((0, 0x10), (1, 0x20)), hp=(16, 16), used_cores=(), used_gpus=(), nic_used=(), busy_time=0.0,
              maintenance=False, groups=None, active=True):
    lab = {NFD + "nfd-extras-cpu.numSockets": str(sockets), NFD + "nfd-extras-cpu.num_cores": str(phys)}
    if smt:
        lab[NFD + "cpu-hardware_multithreading"] = "true"
    for j, (numa, mbps, sw) in enumerate(nics):
        lab[NFD + f"nfd-extras-nic.eth{j}.mlx.{0xC0FFEE00 + j:012x}.{mbps}Mbs.{numa}.{sw:x}.{j}.0"] = "true"
    for g, (numa, sw) in enumerate(gpus):
        lab[NFD + f"nfd-extras-gpu.{g}.V100.{numa}.{sw:x}"] = "true"
    lab["DATA_PLANE_VLAN"] = "9"
    lab["DATA_DEFAULT_GW"] = "10.9.0.1/32"
    if groups:
        lab["NHD_GROUP"] = ".".join(groups)
    if maintenance:
        lab[refmodel.MAINT_LABEL] = "yes"
    return dict(name=name, labels=lab, hugepages=list(hp), active=active, used_cores=list(used_cores),
                used_gpus=list(used_gpus), nic_pods_used=list(nic_used) or [0] * len(nics), busy_time=busy_time)


def pod(groups, map_type="NUMA", hp=0, misc=0, misc_smt=False, pod_groups=("default",)):
    return {"spec": dict(map_type=map_type, hugepages_gb=hp, misc=misc, misc_smt=misc_smt, groups=groups),
            "groups": list(pod_groups)}


def grp(proc=2, helpers=0, rx=0, tx=0, gpus=(), proc_smt=False, helper_smt=False):
    return dict(proc=proc, helpers=helpers, rx=rx, tx=tx, gpus=list(gpus), proc_smt=proc_smt, helper_smt=helper_smt)


def handcrafted(ref):
    """The hand-derived behaviours listed in SURVEY.md §8c, pinned by the reference itself."""
    clock = 1.0e6
    nodes = [
        node_desc("a-gpu"),                                                   # plain 2-GPU node
        node_desc("b-nogpu", gpus=()),                                        # GPU-less node (preferred by CPU pods)
        node_desc("c-busy", busy_time=clock - 5.0),                           # deployed to 5 s ago
        node_desc("d-nonic", nics=()),                                        # no schedulable NIC at all
        node_desc("e-slownic", nics=((0, 10000, 0x10), (1, 10000, 0x20))),   # NICs under the 11 Gb/s threshold
        node_desc("f-hp4", hp=(16, 4), gpus=()),                              # 4 free hugepages
        node_desc("g-maint", maintenance=True, gpus=()),
        node_desc("h-usednic", nic_used=(1, 1), gpus=()),                     # both NICs claimed by other pods
        node_desc("i-onesock", sockets=1, phys=8, nics=((0, 25000, 0x10),), gpus=((0, 0x10),)),
        node_desc("j-56g", nics=((0, 56000, 0x10), (0, 25000, 0x11), (1, 40000, 0x20)), gpus=((0, 0x10), (0, 0x11), (1, 0x20))),
        node_desc("k-halfsmt", used_cores=[32 + c for c in range(0, 12)], gpus=()),   # thread-1 of socket 0 cores in use
        node_desc("l-other", groups=["edge", "lab"], gpus=()),
        node_desc("m-inactive", active=False, gpus=()),
        node_desc("n-gpufull", used_gpus=(0, 1)),
    ]
    pods = [
        pod([grp(4, 1), grp(4, 1)], misc=2),                                  # CPU-only, 2 groups  -> prefers b-nogpu
        pod([grp(3, gpus=(1,))]),                                             # GPU pod -> busy node skipped
        pod([grp(2)], map_type="PCI"),                                        # PCI with 0 GPUs (quirk Q2)
        pod([grp(2, gpus=(0,))], map_type="PCI"),
        pod([grp(2)], hp=4), pod([grp(2)], hp=5), pod([grp(2)], hp=16), pod([grp(2)], hp=17),   # strict > (Q6)
        pod([grp(2)], map_type="NONE"),
        pod([grp(2, rx=90, tx=90)]), pod([grp(2, rx=90.00000000000001)]), pod([grp(2, rx=50.4)]),
        pod([grp(2, rx=50.400000000000006)]), pod([grp(2, rx=50.40000000000001)]),
        pod([grp(2, rx=40, tx=1), grp(2, rx=40, tx=1), grp(2, rx=10, tx=1)]),                       # three groups
        pod([grp(2, rx=30), grp(2, rx=30), grp(2, rx=30.1)]),
        pod([grp(2, rx=0.1), grp(2, rx=0.2), grp(2, rx=0.3)]),
        pod([grp(7, 2, proc_smt=True, helper_smt=True)], misc=3, misc_smt=False),                   # Q1 misc halving
        pod([grp(7, 2)], misc=3, misc_smt=True),
        pod([grp(8), grp(8)], misc=1),
        pod([grp(2, gpus=(1, 1))]), pod([grp(2, gpus=(1,)), grp(2, gpus=(1,))]),
        pod([grp(2, gpus=(1,)), grp(2, gpus=(1,))], map_type="PCI"),
        pod([grp(2, gpus=(1,)), grp(2, gpus=(1,)), grp(2, gpus=(1,))], map_type="PCI"),
        pod([grp(2)], pod_groups=("edge",)), pod([grp(2)], pod_groups=("lab", "default")),
        pod([grp(2)], pod_groups=("nowhere",)),
        pod([grp(30)]), pod([grp(6), grp(6), grp(6)], misc=2),
    ]
    return run_case(ref, clock, nodes, pods)


def beyond_layout(ref):
    """tests/golden/beyond/beyond_layout.json: a cluster that holds two nodes the device layout cannot mirror - four sockets; two
    sockets of 96 physical cores - among ordinary ones.  The product answers for the other nodes exactly as the reference does
    for THEM and names the two (FindNode never raises): `expected` is the reference on the
    cluster without the two, `expected_whole` the reference on the whole cluster (where they differ the reference chose one of the
    two), `feasible` the reference's per-node verdicts on the whole cluster."""
    clock = 2.0e6
    rng = np.random.default_rng(4242)
    nodes = util.random_cluster_desc(4243, 14)
    odd = [node_desc("quad-socket", sockets=4, phys=64, nics=((0, 100000, 0x10), (1, 100000, 0x20), (2, 100000, 0x30), (3, 100000, 0x40)),
                     gpus=((0, 0x10), (3, 0x40))),
           node_desc("wide-socket", sockets=2, phys=192, gpus=((0, 0x10), (1, 0x20)))]
    nodes = nodes[:4] + [odd[0]] + nodes[4:9] + [odd[1]] + nodes[9:]
    pods = [{"spec": util.random_pod_spec(rng), "groups": ["default"]} for _ in range(30)]
    pods += [pod([grp(40)]), pod([grp(20), grp(20)], misc=2), pod([grp(70, proc_smt=True)])]     # only the wide node has that many cores per socket
    whole = run_case(ref, clock, nodes, pods)
    rest = run_case(ref, clock, [d for d in nodes if d["name"] not in ("quad-socket", "wide-socket")], pods)
    whole["expected_whole"] = whole["expected"]
    whole["expected"] = rest["expected"]
    whole["unmirrored"] = ["quad-socket", "wide-socket"]
    return whole


def main():
    ref = ref_loader.load()
    os.makedirs(OUT, exist_ok=True)
    os.makedirs(os.path.join(OUT, "beyond"), exist_ok=True)
    case = beyond_layout(ref)
    with open(os.path.join(OUT, "beyond", "beyond_layout.json"), "w") as f:
        json.dump(case, f, separators=(",", ":"))
    print("beyond/beyond_layout:", len(case["nodes"]), "nodes x", len(case["pods"]), "pods,",
          sum(1 for a, b in zip(case["expected"], case["expected_whole"]) if a != b), "pods the reference places on a node beyond the layout")
    cases = {"handcrafted": handcrafted(ref)}
    for seed in range(4):
        rng = np.random.default_rng(9000 + seed)
        descs = util.random_cluster_desc(7000 + seed, 20)
        pods = [{"spec": util.random_pod_spec(rng), "groups": list(rng.choice(["default", "alpha", "beta"],
                 size=int(rng.integers(1, 3)), replace=False))} for _ in range(24)]
        cases[f"random{seed}"] = run_case(ref, util.CLOCK, descs, pods)
    for cfg in (1, 2, 3, 4, 5):
        spec = synth.make_cluster(cfg, n_nodes=32 if cfg == 1 else 96)
        ps, pg = synth.make_pods(cfg, n_pods=1 if cfg == 1 else 20)
        cases[f"synth_c{cfg}"] = run_case(ref, spec.clock_now, [spec.describe(i) for i in range(spec.n)],
                                          [{"spec": s, "groups": g} for s, g in zip(ps, pg)])
    for name, case in cases.items():
        path = os.path.join(OUT, name + ".json")
        with open(path, "w") as f:
            json.dump(case, f, separators=(",", ":"))
        placed = sum(1 for e in case["expected"] if e[0] is not None)
        print(f"{name}: {len(case['nodes'])} nodes x {len(case['pods'])} pods, {placed} placed, "
              f"{os.path.getsize(path) // 1024} KiB")


if __name__ == "__main__":
    main()
