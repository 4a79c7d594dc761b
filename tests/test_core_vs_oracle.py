"""The kernels' shared arithmetic (fit_core.h / winner_map.h, host build) + the packer against the
oracle and the golden vectors.  CPU only: this is what lets the HIP path be right first time."""
import glob
import json
import os

import numpy as np
import pytest

from nhd_amd import pack
from workload import planes, refmodel, synth
from oracle import nhd_oracle as O
from tests import harness, util

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.json")))


def decode(packer, table, reqs, score, maps, names):
    out = []
    for p in range(len(reqs)):
        s = int(score[p])
        if s == 0:
            out.append([None])
            continue
        idx = 0x7FFFFFFFFFFFFFFF - (s & 0x7FFFFFFFFFFFFFFF)
        G = int(reqs[p]["n_groups"])
        m = maps[p]
        assert m["valid"] == 1
        out.append([names[idx], {"gpu": [int(x) for x in m["gpu"][:G]], "cpu": [int(x) for x in m["cpu"][:G + 1]],
                                 "nic": [[int(a), int(b)] for a, b in zip(m["nic_numa"][:G], m["nic_idx"][:G])]}])
    return out


def bitmap_rows(bitmap, n, P):
    rows = []
    for p in range(P):
        rows.append("".join("1" if int(bitmap[i // 64, p]) >> (i % 64) & 1 else "0" for i in range(n)))
    return rows


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-5] for p in GOLDEN])
def test_core_reproduces_golden(path):
    with open(path) as f:
        case = json.load(f)
    nl = util.build_cluster(case["nodes"])
    pk = pack.Packer()
    table = pk.pack_nodes(nl)
    tops = [refmodel.make_topology(p["spec"]) for p in case["pods"]]
    reqs = pk.digest_many(tops, [p["groups"] for p in case["pods"]])
    score, bitmap, maps = harness.find(pk, table, reqs, case["clock"])
    assert bitmap_rows(bitmap, table.n, len(reqs)) == case["feasible"]
    assert decode(pk, table, reqs, score, maps, table.names) == case["expected"]


@pytest.mark.parametrize("seed", range(20))
def test_core_matches_oracle_random(seed):
    nl = util.random_cluster(31000 + seed, 40)
    rng = np.random.default_rng(seed)
    specs = [util.random_pod_spec(rng, max_groups=4 if seed % 4 == 0 else 3) for _ in range(30)]
    tops = [refmodel.make_topology(s) for s in specs]
    pk = pack.Packer()
    table = pk.pack_nodes(nl)
    reqs = pk.digest_many(tops)
    score, bitmap, maps = harness.find(pk, table, reqs, util.CLOCK)
    got = decode(pk, table, reqs, score, maps, table.names)
    rows = bitmap_rows(bitmap, table.n, len(reqs))
    for p, top in enumerate(tops):
        want = O.find_node(nl, top, util.CLOCK)
        want = [None] if want[0] is None else [want[0], {"gpu": list(want[1]["gpu"]), "cpu": list(want[1]["cpu"]),
                                                          "nic": [list(x) for x in want[1]["nic"]]}]
        assert got[p] == want, specs[p]
        assert rows[p] == "".join("1" if O.feasible(v, top, util.CLOCK) else "0" for v in nl.values())


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5])
def test_spec_route_equals_object_route(cfg):
    spec = synth.make_cluster(cfg, n_nodes=200)
    pk_a, pk_b = pack.Packer(), pack.Packer()
    ta = pk_a.pack_nodes(spec.build_nodes())
    tb = planes.planes_from_spec(pk_b, spec)
    for f in ("p0", "p1", "p2"):
        assert np.array_equal(getattr(ta, f), getattr(tb, f)), f
    assert np.array_equal(ta.p4["busy_time"], tb.p4["busy_time"])
    def names(pk, bits):
        return [frozenset(pk.group_names[k] for k in range(64) if int(b) >> k & 1) for b in bits]
    assert names(pk_a, ta.p3["groups"]) == names(pk_b, tb.p3["groups"])
    assert names(pk_a, np.asarray(pk_a.group_sets, dtype=object)[ta.p4["group_set"]]) == \
           names(pk_b, np.asarray(pk_b.group_sets, dtype=object)[tb.p4["group_set"]])
    assert pack.resolve_signatures(pk_a, ta) == pack.resolve_signatures(pk_b, tb)
    for f in ("nic_cnt", "sw_free", "nic_sw", "numa_nodes", "nic_pods"):
        assert np.array_equal(ta.detail[f], tb.detail[f]), f
    for f in ("t0", "t1", "hp_total"):                        # what ResetResources goes back to
        assert np.array_equal(ta.origin[f], tb.origin[f]), f
    live = np.arange(pack.MAX_NICS_PER_NUMA)[None, None, :] < ta.detail["nic_cnt"][:, :, None]
    ca = np.where(live, np.asarray(pk_a.caps)[ta.detail["nic_cls"]], -1.0)
    cb = np.where(live, np.asarray(pk_b.caps)[tb.detail["nic_cls"]], -1.0)
    assert np.array_equal(ca, cb)
    live = np.arange(pack.MAX_NICS_PER_NUMA)[None, None, :] < ta.detail["nic_cnt"][:, :, None]
    assert np.array_equal(np.where(live, np.asarray(pk_a.caps)[ta.origin["nic_base"]], -1.0),
                          np.where(live, np.asarray(pk_b.caps)[tb.origin["nic_base"]], -1.0))


def test_candidate_mask_and_sharding_agree():
    spec = synth.make_cluster(4, n_nodes=300)
    pods, groups = synth.make_pods(4, n_pods=70)
    pk = pack.Packer()
    table = planes.planes_from_spec(pk, spec)
    reqs = pk.digest_many([refmodel.make_topology(s) for s in pods], groups)
    full, bm, _ = harness.find(pk, table, reqs, spec.clock_now, want_map=False)
    parts = []
    for lo, hi in ((0, 128), (128, 300)):
        s, _, _ = harness.find(pk, table.slice(lo, hi), reqs, spec.clock_now, global_base=lo, want_map=False)
        parts.append(s)
    assert np.array_equal(np.maximum(parts[0], parts[1]), full)
    cand = np.zeros(bm.shape[0], np.uint64)
    cand[1:] = np.uint64(0xFFFFFFFFFFFFFFFF)          # forbid the first 64 nodes
    masked, bm2, _ = harness.find(pk, table, reqs, spec.clock_now, cand=cand, want_map=False)
    assert np.all(bm2[0] == 0) and np.array_equal(bm2[1:], bm[1:])


def test_register_and_generic_set_models_agree():
    spec = synth.make_cluster(4, n_nodes=400)
    pods, groups = synth.make_pods(4, n_pods=128)
    pk = pack.Packer()
    table = planes.planes_from_spec(pk, spec)
    reqs = pk.digest_many([refmodel.make_topology(s) for s in pods], groups)
    a = harness.find(pk, table, reqs, spec.clock_now, force_generic=False)
    b = harness.find(pk, table, reqs, spec.clock_now, force_generic=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and a[2]["valid"].sum() > 100


def test_hot_and_cold_table_sections_agree_on_heterogeneous_clusters():
    """The fit role reads a node's GPU / NIC verdict from the X row of its interned (free GPUs, signatures) class; the
    mapping roles and mode B read the per-signature cold rows.  harness.find() asserts, for every (node, tile) pair it
    evaluates, that both give the same word and that the threshold form of Node.IsBusy equals the subtraction form
    (clocks on both sides of the busy window)."""
    for cfg, n in ((3, 1500), (5, 1000), (2, 500)):
        spec = synth.make_cluster(cfg, n_nodes=n)
        pods, groups = synth.make_pods(cfg, n_pods=96)
        pk = pack.Packer()
        t = planes.planes_from_spec(pk, spec)
        reqs = pk.digest_many([refmodel.make_topology(s) for s in pods], groups)
        for now in (spec.clock_now, spec.clock_now + 24.9, spec.clock_now + 25.1, spec.clock_now + 1e6):
            harness.find(pk, t, reqs, now, want_map=False)
    nl = util.random_cluster(77, 300)
    pk = pack.Packer()
    t = pk.pack_nodes(nl)
    rng = np.random.default_rng(5)
    reqs = pk.digest_many([refmodel.make_topology(util.random_pod_spec(rng, max_groups=4)) for _ in range(70)])
    harness.find(pk, t, reqs, util.CLOCK, want_map=False)


def test_digest_forms_agree_on_a_closed_config5_dictionary():
    """The digest's signature rows by pool type (dict_stream.h: (type, multiplicity) pairs, k-fold unions per type) and the fit
    role's pair rows against the plain forms, on config 5's dictionary closed under claims (272 signatures, eight one-NIC pools per
    PCI-mode signature) - the harness counts every disagreement (harness.find asserts zero) - and the typed stream really is in use."""
    import ctypes
    from nhd_amd import pack
    from workload import planes, refmodel, synth
    from tests import harness
    spec = synth.make_cluster(5, n_nodes=1536)
    pods, groups = synth.make_pods(5, n_pods=150)
    tops = [refmodel.make_topology(s) for s in pods]
    pk = pack.Packer()
    table = planes.planes_from_spec(pk, spec)
    reqs = pk.digest_many(tops, groups)
    before = len(pk.sigs)
    pk.close_signatures()
    assert len(pk.sigs) > before + 50                           # the closure added states no node is in
    words, ntypes = harness.typed_stream(pk)
    assert 0 < words <= 6144 and ntypes >= 4, (words, ntypes)   # in use on the device: the stream fits the digest block's LDS
    score, _, _ = harness.find(pk, table, reqs, spec.clock_now, want_bitmap=False)
    assert (score != 0).sum() > 50
