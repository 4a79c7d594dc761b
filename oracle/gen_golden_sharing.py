#!/usr/bin/env python3
"""Generate tests/golden/sharing/*.json by running the reference (imported from /root/reference) with its module constant
nhd/Node.py:20 ENABLE_SHARING flipped to True - the one edit an operator makes to let pods share a NIC; nothing else of the
reference is touched.  TEST INFRASTRUCTURE; run in the build container only:

    python oracle/gen_golden_sharing.py

GetFreeNumaNicResources then prices a NIC per direction at speed * NIC_BW_AVAIL_PERCENT - speed_used[x] (nhd/Node.py:289-291)
and the commit step's `speed_used[sidx] += nic_speed` (nhd/Node.py:754) decides what later pods still find - pods_used plays no
part.  Fields (those of oracle/gen_golden_big.py):
  snapshot[i]  = Matcher().FindNode over the whole cluster for pod i, every pod against the same state (mode A)
  sequence[i]  = [node, mapping, ids] or [None]: the scheduler's loop (nhd/NHDScheduler.py:274-304)
  final[name]  = the node afterwards, with every NIC's speed_used
Node descriptions carry `nic_speed_used` (traffic already on a NIC) next to `nic_pods_used`; every processing group has one RX and
one TX core (workload/refmodel.make_topology) - but for `sharing_split`, whose groups bring up to three pairs."""
import contextlib
import io
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from workload import refmodel                # noqa: E402
from oracle import ref_loader                # noqa: E402
from oracle.gen_golden_wide import jsonable, ids_of, node_state   # noqa: E402
from tests import util                       # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "sharing")


def state_of(n):
    s = node_state(n)
    s["speed_used"] = [[float(x.speed_used[0]), float(x.speed_used[1])] for x in n.nics]
    return s


def traffic_pod(rng, max_groups):
    s = util.random_pod_spec(rng, max_groups=max_groups)
    for g in s["groups"]:                                     # every group brings traffic: NICs fill up direction by direction
        g["rx"] = float(rng.choice([5, 10, 22.5, 25, 40, 0.1, 33.3]))
        g["tx"] = float(rng.choice([0, 5, 12.25, 25, 45, 0.7]))
        if rng.random() < 0.8:                                # mostly pods without GPUs: they are not stopped by a node's busy time
            g["gpus"] = []                                    # (nhd/Matcher.py:107-111), so several land on one node and meet on its NICs
        g["proc"], g["helpers"] = int(rng.integers(2, 4)), int(rng.integers(0, 2))
    s["hugepages_gb"] = int(rng.choice([0, 0, 1]))
    s["misc_smt"] = True                                      # (keeps the commit away from quirk Q1's raise; tests/golden/commit covers Q1)
    if s["map_type"] == "NONE":
        s["map_type"] = "NUMA"
    return s


def split_pod(rng, max_groups):
    """groups with up to three RX / TX core pairs, every speed a multiple of 2^-20 Gb/s (round 6: the commit adds them to speed_used one by
    one, nhd/Node.py:744-764; the product's request record carries their sums - the same f64 values while all sums are exact)"""
    s = traffic_pod(rng, max_groups)
    speeds = [0, 1, 2.5, 5, 10, 0.25, 12.5]
    for g in s["groups"]:
        g["rx"], g["tx"] = float(rng.choice(speeds)), float(rng.choice(speeds))
        extra = int(rng.integers(0, 3))
        g["more_nic_pairs"] = [(float(rng.choice(speeds)), float(rng.choice(speeds))) for _ in range(extra)]
        g["proc"] = max(g["proc"], 2 + 2 * extra)
    return s


def run_case(ref, descs, specs, fname):
    clock = ref_loader.VirtualClock(util.CLOCK).install()
    nl = util.build_cluster(descs, ref)
    snapshot = [jsonable(ref_loader.find_node(nl, refmodel.make_topology(s, ref))) for s in specs]
    sequence = []
    for s in specs:
        top = refmodel.make_topology(s, ref)
        res = ref_loader.find_node(nl, top)
        if res[0] is None:
            sequence.append([None])
            continue
        n = nl[res[0]]
        n.SetBusy()
        with contextlib.redirect_stdout(io.StringIO()):
            nic_list = n.SetPhysicalIdsFromMapping(res[1], top)      # (an IndexError here would make the fixture undefined: let it raise)
        n.ClaimPodNICResources(list({x[0] for x in nic_list}))
        sequence.append(jsonable(res) + [ids_of(top, n)])
    case = {"clock": clock.t, "nodes": descs, "pods": specs, "snapshot": snapshot, "sequence": sequence,
            "final": {name: state_of(n) for name, n in nl.items()}}
    with open(os.path.join(OUT, fname + ".json"), "w") as f:
        json.dump(case, f, separators=(",", ":"))
    shared = sum(1 for st in case["final"].values() for u in st["speed_used"] if u[0] > 0 or u[1] > 0)
    per_node = {}
    for r in sequence:
        if r[0] is not None:
            per_node[r[0]] = per_node.get(r[0], 0) + 1
    print("   pods per node in sequence:", sorted(per_node.values(), reverse=True))
    print(fname, len(descs), "nodes,", len(specs), "pods;", sum(1 for r in snapshot if r[0] is not None), "placed in the snapshot,",
          sum(1 for r in sequence if r[0] is not None), "in sequence;", shared, "NICs carry traffic afterwards", flush=True)


def main():
    ref = ref_loader.load()
    assert ref.node_mod.ENABLE_SHARING is False
    ref.node_mod.ENABLE_SHARING = True                        # nhd/Node.py:20
    try:
        os.makedirs(OUT, exist_ok=True)
        for seed, n_nodes, n_pods, max_groups, fname in ((83001, 10, 60, 3, "sharing_a"), (83002, 6, 70, 4, "sharing_b"), (83003, 14, 50, 2, "sharing_c")):
            rng = np.random.default_rng(seed)
            descs = util.random_cluster_desc(seed, n_nodes, occupancy=0.08)
            for d in descs:                                   # traffic already on some NICs, pods_used whatever it was (it plays no part)
                nnic = len(d["nic_pods_used"])
                d["nic_speed_used"] = [[float(rng.choice([0, 0, 10, 25, 47.5])), float(rng.choice([0, 0, 5, 45]))] for _ in range(nnic)]
            run_case(ref, descs, [traffic_pod(rng, max_groups) for _ in range(n_pods)], fname)
        rng = np.random.default_rng(83004)
        descs = util.random_cluster_desc(83004, 12, occupancy=0.06)
        for d in descs:
            d["nic_speed_used"] = [[float(rng.choice([0, 0, 10, 12.5, 22.5])), float(rng.choice([0, 0, 5, 15.25]))] for _ in range(len(d["nic_pods_used"]))]
        run_case(ref, descs, [split_pod(rng, 3) for _ in range(60)], "sharing_split")
    finally:
        ref.node_mod.ENABLE_SHARING = False


if __name__ == "__main__":
    main()
