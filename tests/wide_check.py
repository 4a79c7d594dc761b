"""Shared checker for tests/golden/beyond/wide_mixed_*.json (oracle/gen_golden_wide.py, the unmodified reference): clusters that
mix ordinary nodes with nodes beyond the fast layout, under both batch semantics - every pod against one snapshot (FindNode's
answers and the per-node verdicts) and the scheduler's loop (FindNode + commit, pod after pod: node, mapping, the physical ids the
reference wrote into the pod's topology, and every node's state afterwards)."""
import glob
import json
import os

import numpy as np

from nhd_amd import pack
from tests import util
from workload import refmodel

FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "beyond", "wide_mixed_*.json")))


def as_jsonable(res):
    if res[0] is None:
        return [None]
    m = res[1]
    return [res[0], {"gpu": [int(x) for x in m["gpu"]], "cpu": [int(x) for x in m["cpu"]], "nic": [[int(a), int(b)] for a, b in m["nic"]]}]


def mirror_state(m, nl, mirror):
    """What the mirror (a downloaded NodeTable) says about every node, in the fixture's terms."""
    out = {}
    for i, (name, node) in enumerate(nl.items()):
        U, cpp = int(node.sockets), int(node.cores_per_proc)
        n_phys, smt = U * cpp, bool(node.smt_enabled)
        by_pos = {(int(x.numa_node), int(x.idx)): k for k, x in enumerate(node.nics)}
        if mirror.wide and i in mirror.wide:
            w = mirror.wide[i]
            t0 = sum(int(x) << (64 * k) for k, x in enumerate(w["t0"]))
            t1 = sum(int(x) << (64 * k) for k, x in enumerate(w["t1"]))
            free = [c for c in range(n_phys) if t0 >> c & 1] + ([c + n_phys for c in range(n_phys) if t1 >> c & 1] if smt else [])
            pods = [0] * len(node.nics)
            for (u, k), pos in by_pos.items():
                pods[pos] = int(w["nic_pods"][u][k])
            out[name] = {"free_cores": free, "free_gpus": [g for g in range(int(w["n_gpus"])) if int(w["gpu_free"]) >> g & 1],
                         "hp_free": int(w["hp_free"]), "nic_pods": pods, "busy_time": float(w["busy_time"])}
            continue
        t0 = [int(x) for x in mirror.p0[i]["t0"]]
        t1 = [int(x) for x in mirror.p1[i]["t1"]]
        free = [u * cpp + b for u in range(U) for b in range(cpp) if t0[u] >> b & 1]
        if smt:
            free += [n_phys + u * cpp + b for u in range(U) for b in range(cpp) if t1[u] >> b & 1]
        pods = [0] * len(node.nics)
        for (u, k), pos in by_pos.items():
            code = pack.get_pods(mirror.detail[i], u, k)
            pods[pos] = code if code < 4 else code - 8
        out[name] = {"free_cores": free, "free_gpus": [g for g in range(len(node.gpus)) if int(mirror.p2[i]["gpu_free"]) >> g & 1],
                     "hp_free": int(mirror.p2[i]["hp_free"]), "nic_pods": pods, "busy_time": float(mirror.p4[i]["busy_time"])}
    return out


def check(path, make_matcher, unpack_bitmap):
    with open(path) as f:
        case = json.load(f)
    nl = util.build_cluster(case["nodes"])
    tops = [refmodel.make_topology(s) for s in case["pods"]]
    m = make_matcher(case["clock"])
    got = m.FindNodes(nl, tops)
    assert [as_jsonable(r) for r in got] == case["snapshot"]
    assert m.unmirrored == {} and set(m.wide_nodes) <= set(case["drawn_wide"]) and len(m.wide_nodes) >= 5
    reqs = m.packer.digest_many(tops)
    _, bm, _ = m.engine.find(reqs, case["clock"], want_bitmap=True, want_map=False)
    for i, row in enumerate(unpack_bitmap(bm, len(nl))):
        assert "".join(str(int(x)) for x in row) == case["feasible"][i], i
    m.attach(nl)
    seq = m.ScheduleBatch(nl, tops, now=case["clock"], apply=True)
    assert [as_jsonable(r) for r in seq] == [w[:2] for w in case["sequence"]]
    assert m.last_placements == [w[2] if w[0] is not None else None for w in case["sequence"]]
    on_wide = sum(1 for w in case["sequence"] if w[0] is not None and w[0] in m.wide_nodes)
    assert on_wide >= 3, on_wide
    state = mirror_state(m, nl, m.engine.download())
    for name, want in case["final"].items():
        assert state[name] == want, (name, state[name], want)
    return m
