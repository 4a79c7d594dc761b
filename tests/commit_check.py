"""Shared checker for the commit-step fixtures (tests/golden/commit, generated from the unmodified reference by
oracle/gen_golden_commit.py): rebuilds the fixture's cluster and pods, lets a `schedule(packer, table, reqs, now)`
callable decide + commit the batch on packed state, and compares nodes, mappings, physical ids and the final packed
state with what the reference did."""
import glob
import json
import os

import numpy as np

from nhd_amd import pack
from workload import refmodel, synth

FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "commit", "*.json")))


def load(path):
    with open(path) as f:
        return json.load(f)


def build(case):
    spec = synth.make_cluster(case["config"], n_nodes=case["n_nodes"])
    pods, groups = synth.make_pods(case["config"], n_pods=case.get("n_pods_drawn", case["n_pods"]))
    pods, groups = pods[:case["n_pods"]], groups[:case["n_pods"]]
    if case.get("force_misc_smt", True):                       # (commit_q1_*: the pods as drawn - quirk Q1's run-on walk)
        for p in pods:
            p["misc_smt"] = True
    nodes = spec.build_nodes()
    tops = [refmodel.make_topology(p) for p in pods]
    pk = pack.Packer()
    table = pk.pack_nodes(nodes)
    reqs = pk.digest_many(tops, groups)
    pk.close_signatures()
    return spec, nodes, tops, pk, table, reqs


def check(case, nodes, tops, table, reqs, node, maps, places, status, final_table):
    """final_table: the planes after the batch (host twin: modified copy; device: downloaded mirror)."""
    names = table.names
    for i, want in enumerate(case["expected"]):
        if want[0] is None:
            assert node[i] < 0, (i, int(node[i]))
            continue
        assert node[i] >= 0 and names[int(node[i])] == want[0], (i, int(node[i]), want[0])
        assert status[i] == pack.COMMIT_OK, (i, int(status[i]))
        G = int(reqs[i]["n_groups"])
        m = maps[i]
        got_map = {"gpu": [int(x) for x in m["gpu"][:G]], "cpu": [int(x) for x in m["cpu"][:G + 1]],
                   "nic": [[int(a), int(b)] for a, b in zip(m["nic_numa"][:G], m["nic_idx"][:G])]}
        assert got_map == want[1], (i, got_map, want[1])
        n = nodes[want[0]]
        ids = pack.expand_placement(places[i], G, int(n.cores_per_proc), int(n.cores_per_proc) * int(n.sockets),
                                    [int(reqs[i]["gpus"][g]) for g in range(G)])
        assert ids == want[2], (i, ids, want[2])
    for k, name in enumerate(names):
        want = case["final"][name]
        d = final_table.detail[k]
        got = {"t0": [int(x) for x in final_table.p0[k]["t0"]], "t1": [int(x) for x in final_table.p1[k]["t1"]],
               "gpu_free": int(final_table.p2[k]["gpu_free"]), "hp_free": int(final_table.p2[k]["hp_free"]),
               "busy_time": float(final_table.p4[k]["busy_time"]),
               "nic_claimed": [[int(d["nic_cls"][u][j]) == 0 for j in range(int(d["nic_cnt"][u]))] for u in range(2)],
               "sw_free": [int(x) for x in d["sw_free"]]}
        assert got == want, (name, got, want)


def check_signatures(pk, final_table):
    """Every node's plane-3 signature ids must be what the packer derives from its (committed) detail record."""
    for k in range(final_table.n):
        sn, sp = pk.sigs_from_detail(final_table.detail[k])
        assert [int(x) for x in final_table.p3[k]["sig_numa"]] == sn and [int(x) for x in final_table.p3[k]["sig_pci"]] == sp, k
