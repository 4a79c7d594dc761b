"""nhd/Node.py:20 ENABLE_SHARING = True (VERDICT r04 missing #2) on the host build: the reference-generated fixtures
(tests/golden/sharing, oracle/gen_golden_sharing.py), random clusters against the Python oracle with its switch flipped -
ordinary pods and pods of 5..7 groups, snapshot answers and the scheduler's loop with physical ids and the NICs' speed_used
afterwards - and, in the build container, the oracle with its switch flipped against the reference with its constant flipped."""
import copy

import numpy as np
import pytest

from nhd_amd import pack
from nhd_amd.matcher import HipMatcher
from oracle import nhd_oracle as O
from oracle import ref_loader
from tests import harness, sharing_check, util
from tests.wide_check import as_jsonable
from workload import refmodel


def _host(clock):
    return HipMatcher(clock=lambda: clock, engine_factory=harness.HarnessEngine)


@pytest.mark.parametrize("path", sharing_check.FIXTURES, ids=[p.split("/")[-1] for p in sharing_check.FIXTURES])
def test_reference_generated_sharing_fixtures_on_the_host_build(path):
    sharing_check.check(path, _host)


@pytest.fixture
def sharing(monkeypatch):
    monkeypatch.setattr(refmodel, "ENABLE_SHARING", True)
    monkeypatch.setattr(O, "ENABLE_SHARING", True)


def _traffic(rng, max_groups):
    s = util.random_pod_spec(rng, max_groups=max_groups)
    for g in s["groups"]:
        g["rx"] = float(rng.choice([0, 5, 10, 22.5, 25, 40, 0.1, 33.3]))
        g["tx"] = float(rng.choice([0, 5, 12.25, 25, 45, 0.7]))
        if rng.random() < 0.75:
            g["gpus"] = []
    s["misc_smt"] = True
    if s["map_type"] == "NONE":
        s["map_type"] = "NUMA"
    return s


@pytest.mark.parametrize("seed", range(6))
def test_random_clusters_under_sharing_against_the_oracle(sharing, seed):
    rng = np.random.default_rng(9100 + seed)
    descs = util.random_cluster_desc(9100 + seed, 10, occupancy=0.1)
    for d in descs:
        # (now and then more than a NIC's capacity: the reference then finds a negative remainder on it whatever the pod picks,
        #  Matcher.py:267, and the whole node is out)
        d["nic_speed_used"] = [[float(rng.choice([0, 0, 10, 20, 22.5, 47.5], p=[0.3, 0.2, 0.2, 0.15, 0.1, 0.05])),
                                float(rng.choice([0, 0, 5, 15, 89.5], p=[0.3, 0.3, 0.2, 0.15, 0.05]))] for _ in d["nic_pods_used"]]
    tops = [refmodel.make_topology(_traffic(rng, 4)) for _ in range(40)]
    nl = util.build_cluster(descs)
    m = _host(util.CLOCK)
    got = m.FindNodes(nl, tops)
    assert [as_jsonable(r) for r in got] == [as_jsonable(O.find_node(nl, t, util.CLOCK)) for t in tops]
    assert sum(r[0] is not None for r in got) >= 5
    # the scheduler's loop: the oracle's on a copy of the objects, the product's on its mirror
    ids = []
    want = O.schedule_sequence(copy.deepcopy(nl), tops, [None] * len(tops), util.CLOCK, ids_out=ids)
    m.attach(nl)
    seq = m.ScheduleBatch(nl, tops, now=util.CLOCK, apply=True)
    assert [as_jsonable(r) for r in seq] == [as_jsonable(r) for r in want]
    assert m.last_placements == ids
    assert sum(r[0] is not None for r in seq) >= 3


@pytest.mark.parametrize("seed", range(3))
def test_big_pods_under_sharing_against_the_oracle(sharing, seed):
    """Pods of 5..7 processing groups (nhdfit_big_req) meet the same arithmetic: the general path is written once over both request
    forms, and its symmetry pruning of interchangeable NICs compares speed_used as well."""
    rng = np.random.default_rng(9200 + seed)
    descs = util.random_cluster_desc(9200 + seed, 6, occupancy=0.05)
    for d in descs:
        d["nic_speed_used"] = [[float(rng.choice([0, 10, 10, 20])), float(rng.choice([0, 5, 5, 15]))] for _ in d["nic_pods_used"]]
    specs = []
    for _ in range(8):
        groups = [dict(proc=2, helpers=0, rx=float(rng.choice([0, 5, 10, 25])), tx=float(rng.choice([0, 5, 12.25])), proc_smt=bool(rng.random() < 0.5),
                       helper_smt=False, gpus=[]) for _ in range(int(rng.integers(5, 8)))]
        specs.append(dict(map_type=str(rng.choice(["NUMA", "PCI"], p=[0.8, 0.2])), hugepages_gb=0, misc=int(rng.integers(0, 2)), misc_smt=True, groups=groups))
    tops = [refmodel.make_topology(s) for s in specs]
    nl = util.build_cluster(descs)
    m = _host(util.CLOCK)
    got = m.FindNodes(nl, tops)
    assert [as_jsonable(r) for r in got] == [as_jsonable(O.find_node(nl, t, util.CLOCK)) for t in tops]
    ids = []
    want = O.schedule_sequence(copy.deepcopy(nl), tops, [None] * len(tops), util.CLOCK, ids_out=ids)
    m.attach(nl)
    seq = m.ScheduleBatch(nl, tops, now=util.CLOCK, apply=True)
    assert [as_jsonable(r) for r in seq] == [as_jsonable(r) for r in want]
    assert m.last_placements == ids


@pytest.mark.skipif(not ref_loader.available(), reason="reference tree not present")
@pytest.mark.parametrize("seed", range(3))
def test_oracle_with_its_switch_flipped_equals_the_reference_with_its_constant_flipped(sharing, seed):
    """Pins oracle/nhd_oracle.py's ENABLE_SHARING branch (nhd/Node.py:289-291): FindNode + commit, pod after pod, on the reference's
    own objects with nhd.Node.ENABLE_SHARING = True against the oracle on stand-in objects of the same descriptions."""
    import contextlib
    import io
    ref = ref_loader.load()
    rng = np.random.default_rng(9300 + seed)
    descs = util.random_cluster_desc(9300 + seed, 8, occupancy=0.1)
    for d in descs:
        d["nic_speed_used"] = [[float(rng.choice([0, 0, 10, 20, 47.5], p=[0.3, 0.3, 0.2, 0.15, 0.05])), float(rng.choice([0, 5, 15]))] for _ in d["nic_pods_used"]]
    specs = [_traffic(rng, 3) for _ in range(25)]
    ref_loader.VirtualClock(util.CLOCK).install()
    ref.node_mod.ENABLE_SHARING = True
    try:
        rnl = util.build_cluster(descs, ref)
        onl = util.build_cluster(descs)
        placed = 0
        for s in specs:
            rtop, otop = refmodel.make_topology(s, ref), refmodel.make_topology(s)
            want = ref_loader.find_node(rnl, rtop)
            got = O.find_node(onl, otop, util.CLOCK)
            assert as_jsonable(got) == as_jsonable(want), s
            if want[0] is None:
                continue
            n = rnl[want[0]]
            n.SetBusy()
            with contextlib.redirect_stdout(io.StringIO()):
                nic_list = n.SetPhysicalIdsFromMapping(want[1], rtop)
            n.ClaimPodNICResources(list({x[0] for x in nic_list}))
            O.commit(onl[got[0]], otop, got[1], util.CLOCK)
            for a, b in zip(n.nics, onl[got[0]].nics):
                assert [float(x) for x in a.speed_used] == [float(x) for x in b.speed_used]
            placed += 1
        assert placed >= 3
    finally:
        ref.node_mod.ENABLE_SHARING = False


def _odd_traffic(rng):
    s = _traffic(rng, 3)
    for g in s["groups"]:
        g["rx"] = float(rng.choice([-60, -30, -5, 0, 10, 25]))
        g["tx"] = float(rng.choice([-90, -10, 0, 5, 12.25]))
    return s


@pytest.mark.parametrize("seed", range(4))
def test_negative_demands_meet_oversubscribed_nics_as_in_the_reference(sharing, seed):
    """the reference looks for negative remainders AFTER the combination's demands were subtracted (nhd/Matcher.py:262-267),
    so a negative demand can lift a NIC that carries more than its capacity back above zero - and a NIC nobody picks still rules the
    combination out.  Host build against the oracle; in the build container the oracle against the reference with its constant flipped."""
    rng = np.random.default_rng(9400 + seed)
    descs = util.random_cluster_desc(9400 + seed, 8, occupancy=0.05)
    for d in descs:
        d["nic_speed_used"] = [[float(rng.choice([0, 10, 47.5, 95.0], p=[0.4, 0.2, 0.2, 0.2])), float(rng.choice([0, 5, 89.5, 120.0], p=[0.4, 0.2, 0.2, 0.2]))]
                               for _ in d["nic_pods_used"]]
    specs = [_odd_traffic(rng) for _ in range(30)]
    tops = [refmodel.make_topology(s) for s in specs]
    nl = util.build_cluster(descs)
    want = [as_jsonable(O.find_node(nl, t, util.CLOCK)) for t in tops]
    got = _host(util.CLOCK).FindNodes(nl, tops)
    assert [as_jsonable(r) for r in got] == want
    if ref_loader.available():
        ref = ref_loader.load()
        ref_loader.VirtualClock(util.CLOCK).install()
        ref.node_mod.ENABLE_SHARING = True
        try:
            rnl = util.build_cluster(descs, ref)
            assert [as_jsonable(ref_loader.find_node(rnl, refmodel.make_topology(s, ref))) for s in specs] == want
        finally:
            ref.node_mod.ENABLE_SHARING = False


def _split_traffic(rng, exact=True):
    """pods whose groups carry up to three RX / TX core pairs (nhd/Node.py:744-764 adds their speeds to speed_used one by one)"""
    s = util.random_pod_spec(rng, max_groups=3)
    speeds = [0, 0, 1, 2.5, 5, 10, 0.25, 12.5] if exact else [0, 0, 1, 5, 10, 0.1, 3.3, 12.5]
    for g in s["groups"]:
        g["rx"] = float(rng.choice(speeds))
        g["tx"] = float(rng.choice(speeds))
        extra = int(rng.integers(0, 3))
        g["proc"] = max(int(g["proc"]), 2 + 2 * extra)
        g["more_nic_pairs"] = [(float(rng.choice(speeds)), float(rng.choice(speeds))) for _ in range(extra)]
        if rng.random() < 0.75:
            g["gpus"] = []
    s["misc_smt"] = True
    if s["map_type"] == "NONE":
        s["map_type"] = "NUMA"
    return s


@pytest.mark.parametrize("seed", range(4))
def test_groups_with_several_nic_cores_are_exact_while_every_speed_is_dyadic(sharing, seed):
    """VERDICT r05 missing #5: under ENABLE_SHARING the reference's commit adds every RX / TX core's speed to speed_used one after the
    other (nhd/Node.py:744-764); the request record carries a group's sums.  While every value that reaches a speed_used is a
    multiple of 2^-20 (Packer.share_exact) all partial sums are exact and the two are the same f64 value: such pods are answered -
    decisions, ids and every NIC's speed_used afterwards against the oracle, whose commit adds core by core."""
    rng = np.random.default_rng(9500 + seed)
    descs = util.random_cluster_desc(9500 + seed, 16, occupancy=0.05)
    for d in descs:
        d["nic_speed_used"] = [[float(rng.choice([0, 0, 0, 10, 12.5, 22.5])), float(rng.choice([0, 0, 0, 5, 15.25]))] for _ in d["nic_pods_used"]]
    specs = [_split_traffic(rng) for _ in range(40)]
    assert sum(bool(g["more_nic_pairs"]) for s in specs for g in s["groups"]) >= 10
    tops = [refmodel.make_topology(s) for s in specs]
    nl = util.build_cluster(descs)
    m = _host(util.CLOCK)
    got = m.FindNodes(nl, tops)
    assert m.packer.share_exact
    assert [as_jsonable(r) for r in got] == [as_jsonable(O.find_node(nl, t, util.CLOCK)) for t in tops]
    ids = []
    onl = copy.deepcopy(nl)
    want = O.schedule_sequence(onl, tops, [None] * len(tops), util.CLOCK, ids_out=ids)
    m.attach(nl)
    seq = m.ScheduleBatch(nl, tops, now=util.CLOCK, apply=True)
    assert [as_jsonable(r) for r in seq] == [as_jsonable(r) for r in want]
    assert m.last_placements == ids
    assert sum(r[0] is not None for r in seq) >= 5
    share = m.engine.wide_share_download()
    for i, node in enumerate(onl.values()):
        for nic in node.nics:
            if 0 <= nic.numa_node < node.numa_nodes:
                assert [float(share[i]["used"][nic.numa_node][nic.idx][x]) for x in range(2)] == [float(x) for x in nic.speed_used], (node.name, nic.idx)


def test_groups_with_several_nic_cores_are_turned_away_once_a_speed_is_not_dyadic(sharing):
    rng = np.random.default_rng(9600)
    descs = util.random_cluster_desc(9600, 6, occupancy=0.1)
    for d in descs:
        d["nic_speed_used"] = [[0.0, 0.0] for _ in d["nic_pods_used"]]
    nl = util.build_cluster(descs)
    one = dict(map_type="NUMA", hugepages_gb=0, misc=0, misc_smt=True,
               groups=[dict(proc=4, helpers=0, rx=10.0, tx=5.0, more_nic_pairs=[(2.5, 2.5)], gpus=[], proc_smt=False, helper_smt=False)])
    odd = dict(map_type="NUMA", hugepages_gb=0, misc=0, misc_smt=True,
               groups=[dict(proc=2, helpers=0, rx=0.1, tx=5.0, gpus=[], proc_smt=False, helper_smt=False)])
    m = _host(util.CLOCK)
    m.attach(nl)                                                                   # (a stateless call reads every speed_used afresh: nothing carries over)
    t_one, t_odd = refmodel.make_topology(one), refmodel.make_topology(odd)
    assert as_jsonable(m.FindNode(nl, t_one)) == as_jsonable(O.find_node(nl, t_one, util.CLOCK)) and m.FindNode(nl, t_one) != (None,)
    assert m.FindNode(nl, t_odd)[0] is not None and not m.packer.share_exact       # 0.1 Gb/s: sums are no longer exact in any order
    assert m.FindNode(nl, t_one) == (None,)                                        # ... so the group with two RX cores is turned away, loudly
    strict = HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine, strict=True)
    strict.attach(nl)
    strict.FindNode(nl, t_odd)
    with pytest.raises(pack.UnsupportedNode):
        strict.FindNode(nl, t_one)
    m.detach()
    assert m.FindNode(nl, t_one) != (None,)                                        # the objects' own speed_used are all dyadic


@pytest.mark.skipif(not ref_loader.available(), reason="reference tree not present")
@pytest.mark.parametrize("seed", range(2))
def test_oracle_commit_of_several_nic_cores_equals_the_reference(sharing, seed):
    """Pins the oracle's core-by-core speed_used updates (nhd/Node.py:744-764) for groups with several RX / TX cores against the reference
    with its constant flipped - non-dyadic speeds included: the oracle itself has no exactness condition."""
    import contextlib
    import io
    ref = ref_loader.load()
    rng = np.random.default_rng(9700 + seed)
    descs = util.random_cluster_desc(9700 + seed, 8, occupancy=0.1)
    for d in descs:
        d["nic_speed_used"] = [[float(rng.choice([0, 0, 10, 20.1])), float(rng.choice([0, 5, 15]))] for _ in d["nic_pods_used"]]
    specs = [_split_traffic(rng, exact=False) for _ in range(25)]
    ref_loader.VirtualClock(util.CLOCK).install()
    ref.node_mod.ENABLE_SHARING = True
    try:
        rnl = util.build_cluster(descs, ref)
        onl = util.build_cluster(descs)
        placed = 0
        for s in specs:
            rtop, otop = refmodel.make_topology(s, ref), refmodel.make_topology(s)
            want = ref_loader.find_node(rnl, rtop)
            got = O.find_node(onl, otop, util.CLOCK)
            assert as_jsonable(got) == as_jsonable(want), s
            if want[0] is None:
                continue
            n = rnl[want[0]]
            n.SetBusy()
            with contextlib.redirect_stdout(io.StringIO()):
                nic_list = n.SetPhysicalIdsFromMapping(want[1], rtop)
            n.ClaimPodNICResources(list({x[0] for x in nic_list}))
            O.commit(onl[got[0]], otop, got[1], util.CLOCK)
            for a, b in zip(n.nics, onl[got[0]].nics):
                assert [float(x) for x in a.speed_used] == [float(x) for x in b.speed_used]
            placed += 1
        assert placed >= 3
    finally:
        ref.node_mod.ENABLE_SHARING = False


def test_commit_of_a_cached_split_request_is_refused_once_the_regime_is_left(sharing):
    """CommitPlacement reuses the record FindNode digested a moment ago; under ENABLE_SHARING a record with several RX / TX cores per
    group is only good while the mirror's speeds are all dyadic - a value that is not, arriving between the find and the commit,
    makes the commit refuse instead of adding a sum the reference would have accumulated core by core."""
    descs = util.random_cluster_desc(9800, 6, occupancy=0.05)
    for d in descs:
        d["nic_speed_used"] = [[0.0, 0.0] for _ in d["nic_pods_used"]]
    nl = util.build_cluster(descs)
    m = _host(util.CLOCK)
    m.attach(nl)
    split = refmodel.make_topology(dict(map_type="NUMA", hugepages_gb=0, misc=0, misc_smt=True,
                                        groups=[dict(proc=4, helpers=0, rx=10.0, tx=5.0, more_nic_pairs=[(2.5, 2.5)], gpus=[], proc_smt=False, helper_smt=False)]))
    odd = refmodel.make_topology(dict(map_type="NUMA", hugepages_gb=0, misc=0, misc_smt=True,
                                      groups=[dict(proc=2, helpers=0, rx=0.1, tx=5.0, gpus=[], proc_smt=False, helper_smt=False)]))
    r = m.FindNode(nl, split)
    assert r[0] is not None
    ids = m.CommitPlacement(r[0], split, r[1], busy_time=util.CLOCK)                # inside the regime: committed, ids as the oracle's
    rec = {}
    onl = util.build_cluster(descs)
    O.commit(onl[r[0]], split, r[1], util.CLOCK, rec)
    assert ids == rec
    r2 = m.FindNode(nl, split)
    assert r2[0] is not None
    m.FindNodes(nl, [odd, odd])                                                     # (two pods: the one-pod cache keeps the split record)
    assert not m.packer.share_exact
    with pytest.raises(pack.UnsupportedNode):
        m.CommitPlacement(r2[0], split, r2[1], busy_time=util.CLOCK)
