"""Row f2 ("K3 delta update"): release / reclaim / reset and the scheduler's scalar writes mirrored on the device.
Shared by the CPU tests (host-twin engine), the GPU test and the fixture generator (oracle/gen_golden_delta.py, which
runs the same op stream on the UNMODIFIED reference objects).  TEST INFRASTRUCTURE.

A case = a seeded synthetic cluster (workload.synth), a pending list that is scheduled first (FindNode + commit per pod:
the pods' topologies then hold physical ids), then a deterministic stream of operations on the node objects:

    give / take   Node.AddResourcesFromTopology / RemoveResourcesFromTopology of a scheduled pod      nhd/Node.py:587, 530
                  (mostly alternating; now and then the same direction twice - the reference only logs that - which drives
                  pods_used to 2, -1, ... and, rarely, out of the range the packed counters track)
    flag          node.active / node.maintenance                                                   nhd/NHDScheduler.py:533-566
    groups        Node.SetGroups                                                                   nhd/Node.py:308
    busy          node.busy_time
    hugepages     Node.SetHugepages(alloc, free)                                                   nhd/Node.py:489
    reset         Node.ResetResources                                                              nhd/Node.py:144
    find          Matcher.FindNode for a fresh pod on the cluster as it stands

with the state of every node recorded at checkpoints."""
import glob
import json
import os

import numpy as np

from nhd_amd import pack
from workload import refmodel, synth

FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "delta", "*.json")))
CASES = [(3, 40, 90, 260), (4, 32, 110, 300), (5, 48, 140, 340)]          # config, nodes, pods scheduled first, operations
N_FRESH = 24                                                               # pods only used by `find` operations
CHECK_EVERY = 50


def load(path):
    with open(path) as f:
        return json.load(f)


def workload(cfg, n_nodes, n_pods):
    spec = synth.make_cluster(cfg, n_nodes=n_nodes)
    pods, groups = synth.make_pods(cfg, n_pods=n_pods + N_FRESH)
    for p in pods:
        p["misc_smt"] = True                               # the reference's own unwind path is broken
    return spec, pods, groups


def make_ops(seed, names, placed, n_ops, clock):
    """placed: [(pod index, node name)] of the pods the scheduling phase bound."""
    rng = np.random.default_rng(seed)
    held = {i: True for i, _ in placed}
    ops = []
    fresh = 0
    for _ in range(n_ops):
        r = rng.random()
        nm = names[int(rng.integers(len(names)))]
        if r < 0.46 and placed:
            i, node = placed[int(rng.integers(len(placed)))]
            again = rng.random() < 0.12
            give = held[i] != again
            held[i] = not give
            ops.append(["give" if give else "take", node, i])
        elif r < 0.60:
            ops.append(["flag", nm, ["active", "maintenance"][int(rng.integers(2))], bool(rng.integers(2))])
        elif r < 0.67:
            k = int(rng.integers(1, 4))
            ops.append(["groups", nm, ".".join(synth.GROUP_NAMES[int(x)] for x in rng.choice(16, size=k, replace=False))])
        elif r < 0.74:
            ops.append(["busy", nm, clock - float(rng.choice(np.array([2.0, 29.0, 31.0, 500.0])))])
        elif r < 0.80:
            alloc = int(rng.choice(np.array([64, 48])))      # a later reset goes back to THIS total
            ops.append(["hugepages", nm, alloc, int(rng.integers(0, alloc + 1))])
        elif r < 0.85:
            ops.append(["reset", nm])
        else:
            ops.append(["find", None, fresh % N_FRESH])
            fresh += 1
    return ops


def apply_op(nodes, tops, op):
    kind = op[0]
    node = nodes[op[1]]
    if kind == "give":
        node.AddResourcesFromTopology(tops[op[2]])
    elif kind == "take":
        node.RemoveResourcesFromTopology(tops[op[2]])
    elif kind == "flag":
        setattr(node, op[2], op[3])
    elif kind == "groups":
        node.SetGroups(op[2])
    elif kind == "busy":
        node.busy_time = op[2]
    elif kind == "hugepages":
        node.SetHugepages(op[2], op[3])
    elif kind == "reset":
        node.ResetResources()
    else:
        raise ValueError(kind)


def _row(pk, t, i):
    d = t.detail[i]
    nic_cnt = [int(d["nic_cnt"][u]) for u in range(2)]
    return {"t0": [int(x) for x in t.p0[i]["t0"]], "t1": [int(x) for x in t.p1[i]["t1"]],
            "gpu_free": int(t.p2[i]["gpu_free"]), "hp_free": int(t.p2[i]["hp_free"]), "flags": int(t.p2[i]["flags"]),
            "busy_time": float(t.p4[i]["busy_time"]),
            "groups": sorted(pk.group_names[k] for k in range(64) if int(t.p3[i]["groups"]) >> k & 1),
            "nic_claimed": [[int(d["nic_cls"][u][k]) == 0 for k in range(nic_cnt[u])] for u in range(2)],
            "nic_pods": [[pack.get_pods(d, u, k) for k in range(nic_cnt[u])] for u in range(2)],
            "sw_free": [int(x) for x in d["sw_free"]]}


def state_of(nodes):
    """Every node's state in packed terms, from the objects (fresh packer)."""
    pk = pack.Packer()
    t = pk.pack_nodes(nodes)
    return {name: _row(pk, t, i) for i, name in enumerate(t.names)}


def mirror_state(m):
    """The same from the matcher's device mirror (downloaded) - and the signature ids it holds re-derived."""
    t = m.engine.download()
    out = {name: _row(m.packer, t, i) for i, name in enumerate(m._names)}
    for i in range(t.n):
        sn, sp = m.packer.sigs_from_detail(t.detail[i])
        assert [int(x) for x in t.p3[i]["sig_numa"]] == sn and [int(x) for x in t.p3[i]["sig_pci"]] == sp, m._names[i]
        gb = int(t.p3[i]["groups"])
        assert m.packer.group_sets[int(t.p4[i]["group_set"])] == gb
    return out


def as_jsonable(res):
    if res[0] is None:
        return [None]
    return [res[0], {"gpu": list(res[1]["gpu"]), "cpu": list(res[1]["cpu"]), "nic": [list(x) for x in res[1]["nic"]]}]


class Clock:
    def __init__(self, t): self.t = t
    def __call__(self): return self.t


def replay(case, engine_factory=None, devices=None, check=True):
    """The case through HipMatcher in attached mode on stand-in node objects: the scheduling phase as one ScheduleBatch,
    then the op stream through the nodes' own mutators (mirrored as deltas).  Returns (matcher, nodes, binds, finds,
    extra uploads); with check=True every checkpoint of the fixture is asserted on the way (objects AND device mirror)."""
    from nhd_amd.matcher import HipMatcher
    from oracle import nhd_oracle as O
    from tests import sched_standin
    spec, pods, groups = workload(case["config"], case["n_nodes"], case["n_pods"])
    clock = Clock(case["clock"])
    nodes = sched_standin.adopt(spec.build_nodes(), clock)
    tops = [refmodel.make_topology(p) for p in pods]
    P = case["n_pods"]
    m = HipMatcher(clock=clock, engine_factory=engine_factory, devices=devices)
    uploads = []
    orig_upload = m.engine.upload
    m.engine.upload = lambda *a, **k: (uploads.append(a[0].n), orig_upload(*a, **k))[1]
    m.attach(nodes)
    base_uploads = len(uploads)
    binds = sched_standin.check_pending_pods_batched(nodes, m, tops[:P], groups[:P], now=clock.t)
    placed = [(i, b) for i, b in enumerate(binds) if b is not None]
    ops = make_ops(case["seed"], list(nodes), placed, case["n_ops"], case["clock"])
    finds = []
    marks = {c["after"]: c["state"] for c in case.get("checkpoints", [])}
    for k, op in enumerate(ops):
        if op[0] == "find":
            j = P + op[2]
            finds.append(as_jsonable(m.FindNode(O.initial_node_filter(nodes, groups[j]), tops[j])))
        else:
            apply_op(nodes, tops, op)
        if check and k + 1 in marks:
            m.FindNode(nodes, tops[P])                     # any call brings the mirror up to date
            want = marks[k + 1]
            got_obj, got_dev = state_of(nodes), mirror_state(m)
            for name in want:
                assert got_obj[name] == want[name], (k + 1, name, "objects")
                assert got_dev[name] == want[name], (k + 1, name, "device mirror")
    return m, nodes, binds, finds, uploads[base_uploads:]
