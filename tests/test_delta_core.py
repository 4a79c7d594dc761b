"""Row f2 ("K3 delta update", nhd_amd/csrc/commit_core.h apply_delta), host build: release / reclaim / reset and the
scheduler's scalar writes as deltas on the packed state, against the unmodified reference's mutators (build container)
and against the fixtures it generated (tests/golden/delta, oracle/gen_golden_delta.py).  CPU only."""
import contextlib
import io
import os

import numpy as np
import pytest

from nhd_amd import pack
from oracle import nhd_oracle as O
from oracle import ref_loader
from tests import delta_check as D
from tests import harness, sched_standin
from workload import refmodel

needs_ref = pytest.mark.skipif(not ref_loader.available(), reason="reference tree not present")


def _schedule_reference(nodes, tops, groups, P):
    binds = []
    for top, grp in zip(tops[:P], groups[:P]):
        res = ref_loader.find_node(O.initial_node_filter(nodes, grp), top)
        binds.append(res[0])
        if res[0] is None:
            continue
        n = nodes[res[0]]
        n.SetBusy()
        with contextlib.redirect_stdout(io.StringIO()):
            nic_list = n.SetPhysicalIdsFromMapping(res[1], top)
        n.ClaimPodNICResources(list({x[0] for x in nic_list}))
    return binds


def _delta_for(pk, index, node, tops, op):
    kind = op[0]
    if kind in ("give", "take"):
        return pk.delta_from_topology(index, node, tops[op[2]], pack.DELTA_GIVE if kind == "give" else pack.DELTA_TAKE)
    return pk.delta_scalar(index, node, {"flag": "active", "groups": "groups", "busy": "busy_time", "hugepages": "hugepages",
                                         "reset": "reset"}[kind])


@needs_ref
@pytest.mark.parametrize("cfg,n_nodes,n_pods,n_ops", [(3, 24, 60, 400), (4, 20, 70, 400), (5, 30, 80, 400), (2, 16, 40, 200)])
def test_deltas_reproduce_reference_mutators(cfg, n_nodes, n_pods, n_ops):
    """Every operation: the reference's own mutator on its own Node object, the delta on the packed table (host twin of
    k_delta) - the table must equal a fresh pack of the objects after every step.  Statuses: NEW_SIG -> signatures
    interned from the detail record alone; REPACK (a pods_used counter out of range) -> that node re-packed."""
    ref = ref_loader.load()
    spec, pods, groups = D.workload(cfg, n_nodes, n_pods)
    clock = ref_loader.VirtualClock(spec.clock_now).install()
    nodes = spec.build_nodes(ref)
    tops = [refmodel.make_topology(p, ref) for p in pods]
    binds = _schedule_reference(nodes, tops, groups, n_pods)
    placed = [(i, b) for i, b in enumerate(binds) if b is not None]
    assert len(placed) > 10
    pk = pack.Packer()
    table = pk.pack_nodes(nodes)
    index = {nm: i for i, nm in enumerate(table.names)}
    ops = [op for op in D.make_ops(77 + cfg, list(nodes), placed, n_ops, clock.t) if op[0] != "find"]
    seen = {0: 0, 1: 0, 2: 0}
    for k, op in enumerate(ops):
        node = nodes[op[1]]
        with contextlib.redirect_stdout(io.StringIO()):
            D.apply_op(nodes, tops, op)
        d = _delta_for(pk, index[op[1]], node, tops, op)
        st = int(harness.apply_deltas(pk, table, np.array([d]))[0])
        seen[st] += 1
        i = index[op[1]]
        if st == pack.DELTA_NEW_SIG:
            sn, sp = pk.sigs_from_detail(table.detail[i])
            table.p3[i]["sig_numa"], table.p3[i]["sig_pci"] = sn, sp
        elif st == pack.DELTA_REPACK:
            pk.pack_node_into(node, table, i)
        want = pack.empty_table(1)
        pk.pack_node_into(node, want, 0)
        for f in ("p0", "p1", "p2", "p3", "p4", "detail", "origin"):
            assert getattr(table, f)[i].tobytes() == getattr(want, f)[0].tobytes(), (k, op, f, getattr(table, f)[i], getattr(want, f)[0])
    assert seen[0] > n_ops // 2
    # the whole table, once more, against a fresh pack with the same dictionary
    t2 = pk.pack_nodes(nodes)
    for f in ("p0", "p1", "p2", "p3", "p4", "detail", "origin"):
        assert np.array_equal(getattr(table, f), getattr(t2, f)), f


@needs_ref
@pytest.mark.parametrize("cfg", [3, 4, 5])
def test_standin_mutators_match_reference(cfg):
    """The stand-in node's release / reclaim / reset (tests/sched_standin.py, used where the reference is absent) leave the
    same packed state as the reference's on the same op stream."""
    ref = ref_loader.load()
    spec, pods, groups = D.workload(cfg, 24, 60)
    clock = ref_loader.VirtualClock(spec.clock_now).install()
    rnodes = spec.build_nodes(ref)
    rtops = [refmodel.make_topology(p, ref) for p in pods]
    binds = _schedule_reference(rnodes, rtops, groups, 60)
    snodes = sched_standin.adopt(spec.build_nodes(), D.Clock(clock.t))
    stops = [refmodel.make_topology(p) for p in pods]
    for top, grp, b in zip(stops[:60], groups[:60], binds):
        if b is None:
            continue
        res = O.find_node(O.initial_node_filter(snodes, grp), top, clock.t)
        assert res[0] == b
        sched_standin.attempt_scheduling(snodes, None, top, grp, match=res)
    assert D.state_of(snodes) == D.state_of(rnodes)
    placed = [(i, b) for i, b in enumerate(binds) if b is not None]
    for k, op in enumerate(D.make_ops(5 + cfg, list(rnodes), placed, 300, clock.t)):
        if op[0] == "find":
            continue
        with contextlib.redirect_stdout(io.StringIO()):
            D.apply_op(rnodes, rtops, op)
        D.apply_op(snodes, stops, op)
        if k % 25 == 0:
            assert D.state_of(snodes) == D.state_of(rnodes), (k, op)
    assert D.state_of(snodes) == D.state_of(rnodes)


@pytest.mark.parametrize("devices", [None, [0, 1, 2]], ids=["one-engine", "three-shards"])
@pytest.mark.parametrize("path", D.FIXTURES, ids=[os.path.basename(p)[:-5] for p in D.FIXTURES])
def test_attached_matcher_mirrors_mutators_as_deltas(path, devices):
    """Reference-generated fixture through HipMatcher (attached, host-twin engine; also sharded over three of them -
    GroupEngine routes every delta to the shard that owns its node): binds, FindNode results in between, node objects and
    mirror at every checkpoint; the mutators travel as deltas - a node is re-packed only where a delta came back with a
    status."""
    case = D.load(path)
    m, nodes, binds, finds, uploads = D.replay(case, engine_factory=harness.HarnessEngine, devices=devices)
    assert binds == case["binds"]
    assert finds == case["finds"]
    assert m.delta_stats["applied"] > case["n_ops"] // 2
    assert len(uploads) == m.delta_stats["repacked"] and all(n == 1 for n in uploads)
    assert m.delta_stats["repacked"] <= m.delta_stats["applied"] // 10


def test_pods_counter_range_and_sticky_overflow():
    """pods_used counters: -3 .. 3 tracked, anything else -> REPACK and sticky; a claimed NIC frees at <= 0 only."""
    pk = pack.Packer()
    spec, pods, groups = D.workload(3, 4, 4)
    nodes = sched_standin.adopt(spec.build_nodes(), D.Clock(spec.clock_now))
    name = next(nm for nm, n in nodes.items() if len(n.nics) >= 2)
    node = nodes[name]
    for n in node.nics:
        n.pods_used = 0
    table = pk.pack_nodes(nodes)
    pk.close_signatures()
    i = table.names.index(name)
    nic = node.nics[0]
    d = np.zeros((), pack.DELTA)
    d["node"], d["nic_n"] = i, 1
    d["nic"][0] = (nic.numa_node << 4) | nic.idx
    base = int(table.origin[i]["nic_base"][nic.numa_node][nic.idx])
    assert base != 0

    def step(op):
        d["op"] = op
        st = int(harness.apply_deltas(pk, table, np.array([d]))[0])
        det = table.detail[i]
        return st, pack.get_pods(det, nic.numa_node, nic.idx), int(det["nic_cls"][nic.numa_node][nic.idx])

    assert step(pack.DELTA_GIVE) == (pack.DELTA_OK, 7, base)             # -1: "was not in use" - still free
    assert step(pack.DELTA_TAKE) == (pack.DELTA_OK, 0, base)             # back to 0: free (Node.py:292 tests > 0)
    assert step(pack.DELTA_TAKE)[1:] == (1, 0)
    assert step(pack.DELTA_TAKE)[1:] == (2, 0)
    assert step(pack.DELTA_TAKE)[1:] == (3, 0)
    assert step(pack.DELTA_TAKE) == (pack.DELTA_REPACK, pack.PODS_LOST, 0)
    assert step(pack.DELTA_GIVE) == (pack.DELTA_REPACK, pack.PODS_LOST, 0)   # sticky until the host re-packs the node
    r = np.zeros((), pack.DELTA)
    r["node"], r["op"] = i, pack.DELTA_RESET
    assert int(harness.apply_deltas(pk, table, np.array([r]))[0]) in (pack.DELTA_OK, pack.DELTA_NEW_SIG)
    assert pack.get_pods(table.detail[i], nic.numa_node, nic.idx) == 0
