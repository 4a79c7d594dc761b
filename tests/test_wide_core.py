"""The general path (nhd_amd/csrc/wide_core.h: nodes beyond the fast layout - 3 or 4 sockets, 65..128 physical cores per
socket) on the host build: its CPython set model for tuples over range(U) against THIS interpreter, and - through HipMatcher on
the host twin - its verdicts, winners, mappings, commits and mode-B decisions against the Python oracle (real sets, pinned to the
unmodified reference) on random clusters that mix ordinary and wide nodes.  No GPU needed."""
import ctypes
import itertools

import numpy as np
import pytest

from nhd_amd import pack
from nhd_amd.matcher import HipMatcher
from oracle import nhd_oracle as O
from tests import harness, util
from workload import refmodel


def code_of(t, U):
    c = 0
    for d in t:
        c = c * U + d
    return c


def tuple_of(code, length, U):
    out = []
    for _ in range(length):
        out.append(code % U)
        code //= U
    return tuple(reversed(out))


def model_list(codes, length, U):
    L = harness.lib()
    L.hh_wide_set_list.restype = ctypes.c_int
    a = np.asarray(codes, np.int16)
    out = np.zeros(max(1, len(a)), np.int16)
    n = L.hh_wide_set_list(a.ctypes.data_as(ctypes.c_void_p), len(a), ctypes.c_uint32(length), ctypes.c_uint32(U), out.ctypes.data_as(ctypes.c_void_p))
    assert n >= 0
    return [tuple_of(int(c), length, U) for c in out[:n]]


@pytest.mark.parametrize("U", [1, 2, 3, 4])
def test_tuple_hash_over_range_u(U):
    L = harness.lib()
    L.hh_wide_tuple_hash.restype = ctypes.c_uint64
    for length in range(1, 6):
        for t in itertools.product(range(U), repeat=length):
            assert L.hh_wide_tuple_hash(ctypes.c_uint32(code_of(t, U)), ctypes.c_uint32(length), ctypes.c_uint32(U)) == hash(t) & 0xFFFFFFFFFFFFFFFF, t


@pytest.mark.parametrize("U,length", [(3, 1), (3, 2), (3, 3), (3, 4), (3, 5), (4, 1), (4, 2), (4, 3), (4, 4), (4, 5), (2, 5)])
def test_set_iteration_order_equals_the_interpreters(U, length):
    """list(set) for sets of tuples over range(U) filled the way the reference fills them (product order, Matcher.py:116-141,
    206-220) and in arbitrary order (set(list) of Matcher.py:346 re-inserts in another set's order): every growth step of the
    table (8 -> 32 -> 128 -> 512 -> 2 048 slots) is crossed on the way."""
    rng = np.random.default_rng(1000 * U + length)
    every = list(itertools.product(range(U), repeat=length))
    assert model_list([code_of(t, U) for t in every], length, U) == list(set(every))        # the full product, ascending
    for _ in range(60):
        keep = rng.random(len(every)) < rng.random()
        sub = [t for t, k in zip(every, keep) if k]
        s = set()
        for t in sub:
            s.add(t)
        assert model_list([code_of(t, U) for t in sub], length, U) == list(s)
        order = list(sub)
        rng.shuffle(order)
        s2 = set(order)
        assert model_list([code_of(t, U) for t in order], length, U) == list(s2)


@pytest.mark.parametrize("U,length", [(3, 2), (3, 3), (3, 4), (4, 2), (4, 3), (4, 4)])
def test_three_way_intersection_order(U, length):
    """list(set(a) & set(b) & set(c)) (Matcher.py:346): iterate the smaller operand, the right one on ties."""
    L = harness.lib()
    L.hh_wide_isect3.restype = ctypes.c_int
    rng = np.random.default_rng(77 * U + length)
    every = list(itertools.product(range(U), repeat=length))
    for _ in range(80):
        parts = []
        for _k in range(3):
            sub = [t for t in every if rng.random() < 0.35 + 0.6 * rng.random()]
            rng.shuffle(sub)
            parts.append(sub)
        want = list(set(parts[0]) & set(parts[1]) & set(parts[2]))
        arrs = [np.asarray([code_of(t, U) for t in p] or [0], np.int16) for p in parts]
        out = np.zeros(len(every), np.int16)
        n = L.hh_wide_isect3(arrs[0].ctypes.data_as(ctypes.c_void_p), len(parts[0]), arrs[1].ctypes.data_as(ctypes.c_void_p), len(parts[1]),
                             arrs[2].ctypes.data_as(ctypes.c_void_p), len(parts[2]), ctypes.c_uint32(length), ctypes.c_uint32(U),
                             out.ctypes.data_as(ctypes.c_void_p))
        assert [tuple_of(int(c), length, U) for c in out[:n]] == want


def norm(res):
    if res[0] is None:
        return (None,)
    m = res[1]
    return (res[0], {"gpu": tuple(int(x) for x in m["gpu"]), "cpu": tuple(int(x) for x in m["cpu"]), "nic": [(int(a), int(b)) for a, b in m["nic"]]})


def unpack(bm, n):
    chunks, P = bm.shape
    bits = np.unpackbits(bm.view(np.uint8).reshape(chunks, P, 8), axis=2, bitorder="little")
    return bits.transpose(1, 0, 2).reshape(P, chunks * 64)[:, :n]


@pytest.mark.parametrize("seed", range(8))
def test_mixed_clusters_vs_python_oracle(seed):
    """FindNode / FindNodes on clusters that mix ordinary and wide nodes: node AND mapping of every pod, and the verdict of
    every (pod, node) pair, against the oracle's explicit enumeration with real sets."""
    nl = util.mixed_cluster(52000 + seed, 40)
    rng = np.random.default_rng(seed)
    tops = [refmodel.make_topology(util.random_pod_spec(rng, max_groups=4 if seed < 2 else 3)) for _ in range(60)]
    m = HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine)
    got = m.FindNodes(nl, tops)
    assert m.unmirrored == {} and len(m.wide_nodes) >= 5
    want = [norm(O.find_node(nl, t, util.CLOCK)) for t in tops]
    assert [norm(r) for r in got] == want
    assert sum(1 for w in want if w[0] is not None and w[0] in m.wide_nodes) >= 3            # the general path does decide pods here
    reqs = m.packer.digest_many(tops)
    _, bm, _ = m.engine.find(reqs, util.CLOCK, want_bitmap=True, want_map=False)
    rows = unpack(bm, len(nl))
    nodes = list(nl.values())
    for p, top in enumerate(tops[:25]):
        for j, node in enumerate(nodes):
            assert bool(rows[p][j]) == O.feasible(node, top, util.CLOCK), (p, node.name)
    # one by one through the scheduler's own form (filtered dict -> FindNode)
    for top, w in list(zip(tops, want))[:20]:
        assert norm(m.FindNode(nl, top)) == w


@pytest.mark.parametrize("seed", range(6))
def test_mode_b_and_commits_on_mixed_clusters(seed):
    """ScheduleBatch (mode B) with wide nodes in the cluster - decisions, mappings and the physical ids of every placement against
    the oracle's FindNode + commit loop - and the attached mirror after the same placements were applied to the node objects
    with the oracle's commit (the wide records are re-packed, the planes of ordinary nodes follow the commit hooks)."""
    nl = util.mixed_cluster(61000 + seed, 30, occupancy=0.15)
    ref_nl = util.mixed_cluster(61000 + seed, 30, occupancy=0.15)
    rng = np.random.default_rng(seed)
    specs = []
    for _ in range(70):
        s = util.random_pod_spec(rng)
        s["misc_smt"] = True
        if s["map_type"] == "NONE":
            s["map_type"] = "NUMA"
        specs.append(s)
    tops = [refmodel.make_topology(s) for s in specs]
    m = HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine)
    m.attach(nl)
    got = m.ScheduleBatch(nl, tops, now=util.CLOCK)
    want, ids = [], []
    for top in tops:
        res = O.find_node(ref_nl, top, util.CLOCK)
        rec = {}
        if res[0] is not None:
            try:
                O.commit(ref_nl[res[0]], top, res[1], util.CLOCK, rec)
            except O.CommitFailure:
                break
        want.append(norm(res))
        ids.append(rec if res[0] is not None else None)
    k = len(want)
    assert k >= 10 and sum(w[0] is not None for w in want) >= 5
    assert [norm(r) for r in got[:k]] == want
    assert m.last_placements[:k] == ids
    assert sum(1 for w in want if w[0] is not None and w[0] in m.wide_nodes) >= 2
    # apply=False left the mirror alone: the snapshot answers are unchanged
    fresh = HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine)
    assert [norm(r) for r in m.FindNodes(nl, tops[:20])] == [norm(r) for r in fresh.FindNodes(nl, tops[:20])]
    # now place the pods for real on the attached objects, pod by pod as AttemptScheduling does (FindNode -> commit on the object)
    for top, w in zip(tops[:k], want):
        res = m.FindNode(nl, top)
        assert norm(res) == w
        if res[0] is not None:
            O.commit(nl[res[0]], top, res[1], util.CLOCK)
            m.mark_dirty(res[0])


def test_single_commit_on_a_wide_node_returns_the_references_ids():
    nl = util.mixed_cluster(70001, 24, wide_share=1.0, occupancy=0.2)
    ref_nl = util.mixed_cluster(70001, 24, wide_share=1.0, occupancy=0.2)
    rng = np.random.default_rng(3)
    m = HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine)
    m.attach(nl)
    done = on_wide = 0
    for _ in range(60):
        s = util.random_pod_spec(rng)
        s["misc_smt"] = True
        top = refmodel.make_topology(s)
        res = m.FindNode(nl, top)
        want = O.find_node(ref_nl, top, util.CLOCK)
        assert norm(res) == norm(want)
        if res[0] is None:
            continue
        rec = {}
        try:
            O.commit(ref_nl[want[0]], top, want[1], util.CLOCK, rec)
        except O.CommitFailure:
            break
        got = m.CommitPlacement(res[0], top, res[1], busy_time=util.CLOCK)
        assert got == rec, (res[0], got, rec)
        done += 1
        if res[0] in m.wide_nodes:                                      # the device-side record now equals a fresh pack of the oracle's node
            on_wide += 1
            i = m._index[res[0]]
            after = m.engine.download(i, 1).wide[0]
            fresh = m.packer.pack_wide(ref_nl[res[0]])
            for f in ("t0", "t1", "gpu_free", "hp_free", "busy_time", "nic_cls", "nic_pods"):
                assert np.array_equal(after[f], fresh[f]), (res[0], f)
        O.commit(nl[res[0]], top, res[1], util.CLOCK)                   # keep the attached objects in step (their hooks re-pack)
        m.mark_dirty(res[0])
    assert done >= 8 and on_wide >= 4


@pytest.mark.parametrize("path", __import__("tests.wide_check", fromlist=["FIXTURES"]).FIXTURES, ids=lambda p: p.split("/")[-1][:-5])
def test_reference_generated_mixed_clusters(path):
    """tests/golden/beyond/wide_mixed_*.json: the unmodified reference's snapshot answers, per-node verdicts, FindNode + commit
    sequence (node, mapping, physical ids) and final node states on clusters with three- / four-socket and 72-128-core-per-socket nodes."""
    from tests import wide_check
    wide_check.check(path, lambda clock: HipMatcher(clock=lambda: clock, engine_factory=harness.HarnessEngine), unpack)
