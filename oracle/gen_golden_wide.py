#!/usr/bin/env python3
"""Generate tests/golden/beyond/wide_mixed_*.json by running the UNMODIFIED reference (imported from /root/reference).
TEST INFRASTRUCTURE; run in the build container only:

    python oracle/gen_golden_wide.py

Clusters that mix ordinary nodes with nodes beyond the product's fast layout - three and four sockets, up to 128 physical cores
per socket, a PCIe switch seen from two NUMA nodes (tests/util.mixed_cluster_desc) - under the two batch semantics:
  snapshot[i]  = Matcher().FindNode over the whole cluster for pod i, every pod against the same state (mode A)
  feasible[i]  = '0'/'1' per node: the reference places pod i on that node when it is the only candidate
  sequence[i]  = [node, mapping, ids] or [None]: the scheduler's loop (nhd/NHDScheduler.py:274-304) - FindNode, SetBusy,
                 SetPhysicalIdsFromMapping, ClaimPodNICResources, pod after pod under a virtual clock; ids = the physical ids the
                 reference wrote into the pod's CfgTopology.  The sequence ends before the first commit the reference raises on.
  final[name]  = the node afterwards: unused logical core ids, unused GPU positions, free hugepages, pods_used per NIC, busy time.
"""
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
from tests import util                       # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "beyond")
CASES = [(81001, 28, 70, 0.5, 0.15, "wide_mixed_a"), (81002, 36, 90, 0.35, 0.25, "wide_mixed_b"), (81003, 20, 60, 1.0, 0.1, "wide_mixed_c")]


def jsonable(res):
    if res[0] is None:
        return [None]
    m = res[1]
    return [res[0], {"gpu": list(m["gpu"]), "cpu": list(m["cpu"]), "nic": [list(x) for x in m["nic"]]}]


def ids_of(top, node):
    pos = {g.device_id: i for i, g in enumerate(node.gpus)}
    return {"groups": [{"cores": [c.core for g in pg.group_gpus for c in g.cpu_cores] + [c.core for c in pg.proc_cores],
                        "helpers": [c.core for c in pg.misc_cores], "gpus": [pos[g.device_id] for g in pg.group_gpus]} for pg in top.proc_groups],
            "misc": [c.core for c in top.misc_cores]}


def node_state(n):
    return {"free_cores": [c.core for c in n.cores if not c.used], "free_gpus": [i for i, g in enumerate(n.gpus) if not g.used],
            "hp_free": int(n.mem.free_hugepages_gb), "nic_pods": [int(x.pods_used) for x in n.nics], "busy_time": float(n.busy_time)}


def main():
    ref = ref_loader.load()
    os.makedirs(OUT, exist_ok=True)
    for seed, n_nodes, n_pods, share, occ, fname in CASES:
        rng = np.random.default_rng(seed)
        descs = util.mixed_cluster_desc(seed, n_nodes, wide_share=share, occupancy=occ)
        specs = []
        for _ in range(n_pods):
            s = util.random_pod_spec(rng, max_groups=4 if fname.endswith("c") else 3)
            s["misc_smt"] = True                              # (keeps the commit away from quirk Q1's raise; tests/golden/commit covers Q1)
            if s["map_type"] == "NONE":
                s["map_type"] = "NUMA"
            specs.append(s)
        clock = ref_loader.VirtualClock(util.CLOCK).install()
        nl = util.build_cluster(descs, ref)
        snapshot, feas = [], []
        for s in specs:
            top = refmodel.make_topology(s, ref)
            snapshot.append(jsonable(ref_loader.find_node(nl, top)))
            feas.append("".join("1" if ref_loader.find_node({name: node}, top)[0] is not None else "0" for name, node in nl.items()))
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
        if len(sequence) < n_pods:                            # final state = the defined prefix replayed on fresh objects
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
        wide = [d["name"] for d in descs if d["name"].startswith("w")]
        case = {"clock": clock.t, "nodes": descs, "pods": specs, "snapshot": snapshot, "feasible": feas, "sequence": sequence,
                "final": {name: node_state(n) for name, n in nl.items()}, "drawn_wide": wide}
        with open(os.path.join(OUT, fname + ".json"), "w") as f:
            json.dump(case, f, separators=(",", ":"))
        print(fname, len(descs), "nodes;", sum(1 for r in snapshot if r[0] is not None), "of", n_pods, "pods placed in the snapshot;",
              sum(1 for r in sequence if r[0] is not None), "of", len(sequence), "defined pods placed in sequence")


if __name__ == "__main__":
    main()
