"""Pods with 5..8 processing groups - more than the table-driven pass holds - on the host build: the general path for requests
(nhd_amd/csrc/wide_core.h templates over nhdfit_big_req, commit_core.h commit_node_t; the device runs the same headers through
big_kernel.h).  Its set model for the longer tuples against THIS interpreter; its verdicts against the table-driven pass on pods
both can express; and - through HipMatcher on the host twin - FindNode / FindNodes / ScheduleBatch / CommitPlacement against the
Python oracle (real sets, explicit enumeration, pinned to the unmodified reference) and against reference-generated fixtures
(tests/golden/big, oracle/gen_golden_big.py).  No GPU needed."""
import itertools

import numpy as np
import pytest

from nhd_amd import pack
from nhd_amd.matcher import HipMatcher
from oracle import nhd_oracle as O
from tests import big_check, harness, util
from tests.test_wide_core import code_of, model_list, norm
from workload import refmodel


def host_matcher(clock=util.CLOCK):
    return HipMatcher(clock=lambda: clock, engine_factory=harness.HarnessEngine)


def big_spec(rng, lo=5, hi=8):
    groups = []
    for _ in range(int(rng.integers(lo, hi + 1))):
        ng = int(rng.choice([0, 1], p=[0.8, 0.2]))
        groups.append(dict(proc=int(rng.integers(2, 4)), helpers=int(rng.integers(0, 2)),
                           rx=float(rng.choice([0, 0, 5, 10, 22.5, 25, 40, 0.1])), tx=float(rng.choice([0, 0, 5, 10, 12.25, 45])),
                           proc_smt=bool(rng.random() < 0.5), helper_smt=bool(rng.random() < 0.5),
                           gpus=[int(rng.integers(0, 2)) for _ in range(ng)]))
    return dict(map_type=str(rng.choice(["NUMA", "PCI"], p=[0.6, 0.4])), hugepages_gb=int(rng.choice([0, 1, 4])),
                misc=int(rng.integers(0, 3)), misc_smt=True, groups=groups)


@pytest.mark.parametrize("U,length", [(2, 5), (2, 6), (2, 7), (2, 8), (2, 9), (3, 6)])
def test_set_order_of_the_longer_tuples(U, length):
    """list(set) of tuples over range(U) as long as a big request makes them (G and G + 1 elements, G <= 8): filled in product
    order and in arbitrary order, against this interpreter's sets."""
    rng = np.random.default_rng(31 * U + length)
    every = list(itertools.product(range(U), repeat=length))
    assert model_list([code_of(t, U) for t in every], length, U) == list(set(every))
    for _ in range(25):
        keep = rng.random(len(every)) < rng.random()
        sub = [t for t, k in zip(every, keep) if k]
        s = set()
        for t in sub:
            s.add(t)
        assert model_list([code_of(t, U) for t in sub], length, U) == list(s)
        order = list(sub)
        rng.shuffle(order)
        assert model_list([code_of(t, U) for t in order], length, U) == list(set(order))


def big_model_list(codes, length, U):
    import ctypes
    L = harness.lib()
    L.hh_big_set_list.restype = ctypes.c_int
    a = np.asarray(codes, np.int32)
    out = np.zeros(max(1, len(a)), np.int32)
    n = L.hh_big_set_list(a.ctypes.data_as(ctypes.c_void_p), len(a), ctypes.c_uint32(length), ctypes.c_uint32(U), out.ctypes.data_as(ctypes.c_void_p))
    assert n >= 0
    return out[:n]


@pytest.mark.parametrize("U,length", [(3, 8), (3, 9), (4, 7), (4, 8), (4, 9)])
def test_set_order_of_the_largest_tuple_sets(U, length):
    """The sets a big request makes on a three- or four-socket node: up to 4^9 = 262 144 tuples - past the point (50 000 keys)
    where CPython's table growth changes from 4 x used to 2 x used - in product order and for random subsets in random order."""
    import ctypes
    rng = np.random.default_rng(57 * U + length)
    every = list(itertools.product(range(U), repeat=length))
    codes = np.arange(len(every), dtype=np.int32)                       # product order = ascending codes
    weights = U ** np.arange(length - 1, -1, -1)

    def as_codes(tuples):
        return (np.asarray(tuples, np.int64) @ weights).astype(np.int32) if len(tuples) else np.zeros(0, np.int32)
    assert np.array_equal(big_model_list(codes, length, U), as_codes(list(set(every))))
    for _ in range(3):
        keep = np.flatnonzero(rng.random(len(every)) < rng.random())
        rng.shuffle(keep)
        order = [every[i] for i in keep]
        assert np.array_equal(big_model_list(keep.astype(np.int32), length, U), as_codes(list(set(order))))
    L = harness.lib()
    L.hh_table_slots.restype = ctypes.c_uint32
    assert L.hh_table_slots(ctypes.c_uint32(5)) == 32 and L.hh_table_slots(ctypes.c_uint32(4)) == 8


def test_big_record_layout_and_digest():
    rng = np.random.default_rng(5)
    p = pack.Packer()
    for _ in range(50):
        spec = big_spec(rng, 1, 8)
        top = refmodel.make_topology(spec)
        r = p.digest_big(top)
        G = len(spec["groups"])
        assert int(r["n_groups"]) == G and int(r["map_type"]) in (1, 2)
        if G <= pack.MAX_GROUPS:                                       # field for field the ordinary record
            q = p.digest(top)
            for f in ("gpus", "cpu_smt", "cpu_nosmt", "n_proc", "n_help", "rx", "tx"):
                assert list(r[f][:4]) == list(q[f]) and not np.any(r[f][4:]), f
            for f in ("hugepages_gb", "misc_smt", "misc_nosmt", "n_misc", "misc_smt_enabled", "nic_use", "map_type"):
                assert int(r[f]) == int(q[f]), f
            assert int(r["smt_bits"]) & 0xF == int(q["smt_bits"]) & 0xF and int(r["smt_bits"]) >> 8 == int(q["smt_bits"]) >> 4
    with pytest.raises(pack.UnsupportedNode):
        p.digest_big(refmodel.make_topology(dict(big_spec(rng), groups=big_spec(rng, 8, 8)["groups"] + big_spec(rng, 1, 1)["groups"])))


@pytest.mark.parametrize("seed", range(4))
def test_general_path_equals_the_table_pass_on_ordinary_pods(seed):
    """Pods with 1..4 groups digested BOTH ways: the general path's verdict for every (pod, node) pair of a random cluster equals
    the table-driven pass's bitmap, and so do the winners' mappings (two independent evaluations of one predicate)."""
    nl = util.random_cluster(41000 + seed, 48)
    rng = np.random.default_rng(seed)
    tops = [refmodel.make_topology(util.random_pod_spec(rng, max_groups=4)) for _ in range(40)]
    m = host_matcher()
    m.FindNodes(nl, tops[:1])                                           # packs the cluster, sets the dictionary
    reqs = m.packer.digest_many(tops)
    big = np.array([m.packer.digest_big(t) for t in tops], dtype=pack.BIG_REQ)
    score, bm, maps = m.engine.find(reqs, util.CLOCK, want_bitmap=True, want_map=True)
    fits, bscore, exhausted = harness.big_eval(m.packer, m.engine.table, m.engine._wide_records(), big, util.CLOCK)
    assert not exhausted
    from tests.test_wide_core import unpack
    assert np.array_equal(unpack(bm, len(nl)).astype(np.uint8), fits.T)
    assert np.array_equal(score, bscore)
    _, bmaps = m.engine.big_find(big, util.CLOCK)
    for p in np.flatnonzero(score != 0):
        G = int(reqs[p]["n_groups"])
        for f, k in (("gpu", G), ("cpu", G + 1), ("nic_numa", G), ("nic_idx", G)):
            assert list(maps[p][f][:k]) == list(bmaps[p][f][:k]), (p, f)


@pytest.mark.parametrize("seed", range(6))
def test_big_pods_vs_python_oracle(seed):
    """FindNodes on clusters of ordinary (and, for odd seeds, wide) nodes with pods of up to eight groups: node and mapping of
    every pod against the oracle's explicit enumeration with real sets; the scheduler's one-by-one form as well."""
    if seed % 2:
        descs = util.mixed_cluster_desc(43000 + seed, 28, wide_share=0.4, occupancy=0.1)
    else:
        descs = util.random_cluster_desc(43000 + seed, 36, occupancy=0.1)
    for d in descs:                                                     # (the oracle enumerates like the reference: few NICs keep it quick)
        keep, lab = 0, {}
        for k, v in d["labels"].items():
            if "nfd-extras-nic" in k:
                keep += 1
                if keep > 4:
                    continue
            lab[k] = v
        d["labels"] = lab
        d["nic_pods_used"] = d["nic_pods_used"][:sum(1 for k in lab if "nfd-extras-nic" in k and "10000Mbs" not in k.replace("100000Mbs", ""))]
    nl = util.build_cluster(descs)
    rng = np.random.default_rng(seed)
    hi = 6 if seed % 2 else 7          # (the Python oracle enumerates 4^G x 4^(G+1) on a four-socket node: seven groups there are tests/golden/big/big_quad's)
    tops = [refmodel.make_topology(big_spec(rng, 5, hi) if rng.random() < 0.7 else util.random_pod_spec(rng, 4)) for _ in range(24)]
    m = host_matcher()
    got = m.FindNodes(nl, tops)
    want = [norm(O.find_node(nl, t, util.CLOCK)) for t in tops]
    assert [norm(r) for r in got] == want
    assert sum(1 for t, w in zip(tops, want) if len(t.proc_groups) > 4 and w[0] is not None) >= 2
    for top, w in list(zip(tops, want))[:8]:
        assert norm(m.FindNode(nl, top)) == w


@pytest.mark.parametrize("seed", range(4))
def test_mode_b_with_big_pods_vs_python_oracle(seed):
    """ScheduleBatch over a batch that mixes ordinary and big pods: decisions, mappings and physical ids against the oracle's
    FindNode + commit loop; apply=False leaves the mirror as it was; then the same pods pod by pod with the commit mirrored through
    CommitPlacement (nhdfit_big_commit for the big ones)."""
    descs = util.mixed_cluster_desc(47000 + seed, 24, wide_share=0.35 if seed % 2 else 0.0, occupancy=0.1)
    for d in descs:
        keep, lab = 0, {}
        for k, v in d["labels"].items():
            if "nfd-extras-nic" in k:
                keep += 1
                if keep > 4:
                    continue
            lab[k] = v
        d["labels"] = lab
        d["nic_pods_used"] = d["nic_pods_used"][:sum(1 for k in lab if "nfd-extras-nic" in k and "10000Mbs" not in k.replace("100000Mbs", ""))]
    nl, ref_nl = util.build_cluster(descs), util.build_cluster(descs)
    rng = np.random.default_rng(100 + seed)
    specs = []
    for _ in range(26):
        s = big_spec(rng, 5, 7) if rng.random() < 0.6 else util.random_pod_spec(rng)
        s["misc_smt"] = True
        if s["map_type"] == "NONE":
            s["map_type"] = "NUMA"
        specs.append(s)
    tops = [refmodel.make_topology(s) for s in specs]
    m = host_matcher()
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
    assert k >= 8 and sum(1 for t, w in zip(tops, want) if len(t.proc_groups) > 4 and w[0] is not None) >= 2
    assert [norm(r) for r in got[:k]] == want
    assert m.last_placements[:k] == ids
    fresh = host_matcher()
    assert [norm(r) for r in m.FindNodes(nl, tops[:12])] == [norm(r) for r in fresh.FindNodes(nl, tops[:12])]      # apply=False: mirror as before
    # pod by pod as AttemptScheduling does; the reference-side commit is the oracle's, on the attached objects (their hooks fire)
    fresh_ref = util.build_cluster(descs)
    for top, w, want_ids in zip(tops[:k], want, ids):
        res = m.FindNode(nl, top)
        assert norm(res) == w
        if res[0] is not None:
            assert m.CommitPlacement(res[0], top, res[1], busy_time=util.CLOCK) == want_ids
            O.commit(nl[res[0]], top, res[1], util.CLOCK)
            m.mark_dirty(res[0])
    del fresh_ref


@pytest.mark.parametrize("path", big_check.FIXTURES, ids=lambda p: p.split("/")[-1][:-5])
def test_reference_generated_big_pods(path):
    """tests/golden/big/*.json: the unmodified reference's answers for pods with 5..8 processing groups."""
    big_check.check(path, host_matcher)


def test_more_than_eight_groups_stay_unanswered_not_wrong():
    rng = np.random.default_rng(9)
    nl = util.random_cluster(49000, 12)
    spec = big_spec(rng, 8, 8)
    spec["groups"] = spec["groups"] + big_spec(rng, 1, 1)["groups"]
    m = host_matcher()
    assert m.FindNodes(nl, [refmodel.make_topology(spec)]) == [(None,)]
    with pytest.raises(pack.UnsupportedNode):
        HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine, strict=True).FindNodes(nl, [refmodel.make_topology(spec)])


@pytest.mark.parametrize("ndev", [2, 3])
def test_big_pods_on_a_sharded_mirror(ndev):
    """HipMatcher(devices=[...]) (engine.GroupEngine over host-twin shards): pods with 5..7 groups mixed with ordinary ones - FindNodes,
    the one-by-one form with a candidate dict, ScheduleBatch with its physical ids - equal the single-shard answers (score words
    carry the global node index, the owner's mapping is kept, the commit lands on the owner's mirror)."""
    descs = util.mixed_cluster_desc(48000 + ndev, 200, wide_share=0.1, occupancy=0.1)
    for d in descs:
        keep, lab = 0, {}
        for k, v in d["labels"].items():
            if "nfd-extras-nic" in k:
                keep += 1
                if keep > 4:
                    continue
            lab[k] = v
        d["labels"] = lab
        d["nic_pods_used"] = d["nic_pods_used"][:sum(1 for k in lab if "nfd-extras-nic" in k and "10000Mbs" not in k.replace("100000Mbs", ""))]
    nl = util.build_cluster(descs)
    rng = np.random.default_rng(ndev)
    specs = []
    for _ in range(30):
        s = big_spec(rng, 5, 7) if rng.random() < 0.6 else util.random_pod_spec(rng)
        s["misc_smt"] = True
        if s["map_type"] == "NONE":
            s["map_type"] = "NUMA"
        specs.append(s)
    tops = [refmodel.make_topology(s) for s in specs]
    one = host_matcher()
    many = HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine, devices=list(range(ndev)))
    want = one.FindNodes(nl, tops)
    assert many.FindNodes(nl, tops) == want
    assert sum(1 for t, w in zip(tops, want) if len(t.proc_groups) > 4 and w[0] is not None) >= 3
    sub = {k: v for i, (k, v) in enumerate(nl.items()) if i % 3}
    for top in [t for t in tops if len(t.proc_groups) > 4][:6]:
        assert many.FindNode(sub, top) == one.FindNode(sub, top)
    one.attach(nl)
    a = one.ScheduleBatch(nl, tops, now=util.CLOCK)
    ids_a = list(one.last_placements)
    many.attach(nl)
    b = many.ScheduleBatch(nl, tops, now=util.CLOCK)
    assert a == b and ids_a == many.last_placements
    assert len({r[0] for r in a if r[0] is not None}) >= 3


def test_nic_search_budget_fails_the_call_never_the_answer(caplog):
    """A (pod, node) pair whose NIC search runs out of steps (NHDFIT_BIG_NIC_BUDGET on the device; a tiny budget injected into the
    host twin here) makes the call fail: FindNode answers (None,) and says why, strict raises - no verdict is guessed."""
    from nhd_amd._lib import NhdFitError
    rng = np.random.default_rng(4)
    nl = util.random_cluster(49500, 16, occupancy=0.0)
    for _ in range(50):                                                  # a pod some node takes: its NIC stage is reached
        top = refmodel.make_topology(big_spec(rng, 6, 6))
        if O.find_node(nl, top, util.CLOCK)[0] is not None:
            break
    assert O.find_node(nl, top, util.CLOCK)[0] is not None

    class Tight(harness.HarnessEngine):
        nic_budget = 3
    with caplog.at_level("ERROR"):
        assert HipMatcher(clock=lambda: util.CLOCK, engine_factory=Tight).FindNode(nl, top) == (None,)
    assert "search budget" in caplog.text
    with pytest.raises(NhdFitError):
        HipMatcher(clock=lambda: util.CLOCK, engine_factory=Tight, strict=True).FindNode(nl, top)
    assert host_matcher().FindNode(nl, top) == norm(O.find_node(nl, top, util.CLOCK))


def test_big_pods_with_the_initial_node_filter_and_invalid_map_types():
    """pod_groups given: the general path applies InitialNodeFilter itself (active && node groups intersect the pod's,
    nhd/NHDScheduler.py:235-247) - against the oracle's filter followed by its FindNode, and against the scheduler's form (filtered
    dict -> FindNode).  A big pod whose map type is neither NUMA nor PCI matches nothing (nhd/Matcher.py:45-47)."""
    descs = util.random_cluster_desc(49700, 60, occupancy=0.05)
    for d in descs:
        d["nic_pods_used"] = [0] * len(d["nic_pods_used"])
    nl = util.build_cluster(descs)
    rng = np.random.default_rng(12)
    tops, groups = [], []
    for _ in range(30):
        s = big_spec(rng, 5, 6)
        for g in s["groups"]:
            g["rx"] = g["tx"] = 0.0
        tops.append(refmodel.make_topology(s))
        groups.append(list(rng.choice(["default", "alpha", "beta", "nobody"], size=int(rng.integers(1, 3)), replace=False)))
    m = host_matcher()
    got = m.FindNodes(nl, tops, pod_groups=groups)
    want = [norm(O.find_node(O.initial_node_filter(nl, g), t, util.CLOCK)) for t, g in zip(tops, groups)]
    assert [norm(r) for r in got] == want
    assert sum(w[0] is not None for w in want) >= 5 and len({w[0] for w in want}) >= 3
    m.attach(nl)
    for t, g, w in list(zip(tops, groups, want))[:12]:
        assert norm(m.FindNode(O.initial_node_filter(nl, g), t)) == w
    none = big_spec(rng, 6, 6)
    none["map_type"] = "NONE"
    assert m.FindNode(nl, refmodel.make_topology(none)) == (None,)


def test_a_big_pod_beyond_the_big_record_is_answered_none_not_raised(caplog):
    """Six groups of which one asks for 300 cores: no record holds it (255 cores per group) - (None,) with the reason in the log,
    the other pods of the call answered as always; strict raises."""
    rng = np.random.default_rng(21)
    nl = util.random_cluster(49800, 16, occupancy=0.0)
    fat = big_spec(rng, 6, 6)
    fat["groups"][2]["proc"] = 300
    ok = big_spec(rng, 5, 5)
    plain = util.random_pod_spec(rng)
    tops = [refmodel.make_topology(s) for s in (ok, fat, plain)]
    m = host_matcher()
    with caplog.at_level("ERROR"):
        got = m.FindNodes(nl, tops)
    assert got[1] == (None,) and "cannot be expressed" in caplog.text
    assert norm(got[0]) == norm(O.find_node(nl, tops[0], util.CLOCK)) and norm(got[2]) == norm(O.find_node(nl, tops[2], util.CLOCK))
    assert m.ScheduleBatch(nl, tops, now=util.CLOCK)[1] == (None,)
    with pytest.raises(pack.UnsupportedNode):
        HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine, strict=True).FindNodes(nl, tops)
