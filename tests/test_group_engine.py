"""One process, several devices (engine.GroupEngine behind HipMatcher(devices=[...])): shard bounds, candidate masks per
shard, max-reduction of the packed scores, owner's mappings, delta uploads that straddle shards.  CPU: the shards are
host-twin engines (tests/harness) and the reduction is the host one; the RCCL path of the same class runs in
tests/test_gpu_parity.py (one device) and in bench.py at N > 1."""
import numpy as np
import pytest

from nhd_amd import pack
from nhd_amd.matcher import HipMatcher
from oracle import nhd_oracle as O
from tests import harness, util
from workload import refmodel, synth


def jsonable(res):
    if res[0] is None:
        return [None]
    return [res[0], {"gpu": list(res[1]["gpu"]), "cpu": list(res[1]["cpu"]), "nic": [list(x) for x in res[1]["nic"]]}]


@pytest.mark.parametrize("ndev", [1, 2, 3, 8])
@pytest.mark.parametrize("cfg,n", [(3, 200), (4, 333), (5, 130)])
def test_sharded_matcher_equals_single_shard(cfg, n, ndev):
    spec = synth.make_cluster(cfg, n_nodes=n)
    nl = spec.build_nodes()
    pods, groups = synth.make_pods(cfg, n_pods=90)
    tops = [refmodel.make_topology(s) for s in pods]
    one = HipMatcher(clock=lambda: spec.clock_now, engine_factory=harness.HarnessEngine)
    many = HipMatcher(clock=lambda: spec.clock_now, engine_factory=harness.HarnessEngine, devices=list(range(ndev)))
    want = one.FindNodes(nl, tops, pod_groups=groups)
    got = many.FindNodes(nl, tops, pod_groups=groups)
    assert got == want
    assert sum(r[0] is not None for r in want) > 10
    # the scheduler's form: filtered candidate dict per pod (a candidate mask per shard)
    for top, grp, w in list(zip(tops, groups, want))[:25]:
        sub = O.initial_node_filter(nl, grp)
        assert many.FindNode(sub, top) == one.FindNode(sub, top)


def test_sharded_attach_tracks_mutations_across_shards():
    spec = synth.make_cluster(3, n_nodes=300)
    nl = spec.build_nodes()
    pods, _ = synth.make_pods(3, n_pods=40)
    tops = [refmodel.make_topology(s) for s in pods]
    m = HipMatcher(clock=lambda: spec.clock_now, engine_factory=harness.HarnessEngine, devices=[0, 1, 2])
    m.attach(nl)
    first = m.FindNodes(nl, tops)
    names = list(nl)
    for nm in (names[0], names[127], names[128], names[129], names[299]):       # both sides of the shard cuts (128-node blocks)
        for c in nl[nm].cores:
            c.used = True
        m.mark_dirty(nm)
    nl[names[5]].maintenance = True
    second = m.FindNodes(nl, tops)
    fresh = HipMatcher(clock=lambda: spec.clock_now, engine_factory=harness.HarnessEngine).FindNodes(nl, tops)
    assert second == fresh
    assert second != first
    sub = {k: v for i, (k, v) in enumerate(nl.items()) if i % 3}
    for top in tops[:10]:
        assert jsonable(m.FindNode(sub, top)) == jsonable(O.find_node(sub, top, spec.clock_now))


def test_more_devices_than_chunks():
    nl = util.random_cluster(9, 70)                       # 2 chunks of 64 nodes, 8 devices: six shards are empty
    rng = np.random.default_rng(2)
    tops = [refmodel.make_topology(util.random_pod_spec(rng)) for _ in range(30)]
    m = HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine, devices=list(range(8)))
    for top in tops:
        assert jsonable(m.FindNode(nl, top)) == jsonable(O.find_node(nl, top, util.CLOCK))


@pytest.mark.parametrize("ndev", [2, 3, 5])
@pytest.mark.parametrize("cfg,n,P", [(3, 200, 260), (4, 150, 220), (5, 130, 200)])
def test_sharded_sequential_batch_equals_the_one_by_one_loop(cfg, n, P, ndev):
    """Mode B across shards (GroupEngine.schedule_batch): GPU-less pods first walk the shards' GPU-less nodes, the rest
    walks all nodes shard by shard - node, mapping and physical ids per pod equal the oracle's pod-by-pod loop over the
    whole cluster (pinned to the reference), and so does the mirror afterwards (apply) / nothing changed (no apply)."""
    spec = synth.make_cluster(cfg, n_nodes=n)
    pods, groups = synth.make_pods(cfg, n_pods=P)
    for p in pods:
        p["misc_smt"] = True
    nl = spec.build_nodes()
    tops = [refmodel.make_topology(s) for s in pods]
    many = HipMatcher(clock=lambda: spec.clock_now, engine_factory=harness.HarnessEngine, devices=list(range(ndev)))
    many.attach(nl)
    before = many.engine.download()
    got = many.ScheduleBatch(nl, tops, pod_groups=groups)                   # apply=False: the shards are put back
    after = many.engine.download()
    for f in ("p0", "p1", "p2", "p3", "p4", "detail"):
        assert np.array_equal(getattr(before, f), getattr(after, f)), f
    ids = []
    ora = spec.build_nodes()
    want = O.schedule_sequence(ora, tops, groups, spec.clock_now, ids_out=ids)
    assert [jsonable(r) for r in got] == [jsonable(w) for w in want]
    assert [p for p in many.last_placements] == ids
    placed = [w for w in want if w[0] is not None]
    assert len(placed) > 40 and len({w[0] for w in placed}) > 10
    # pods that went past the first shard: the hand-over between shards is exercised
    names = list(nl)
    cut = many.engine._bounds[0][1]
    assert any(names.index(w[0]) >= cut for w in placed)
    # apply=True: the mirror holds the committed state of the oracle's node objects
    got2 = many.ScheduleBatch(nl, tops, pod_groups=groups, apply=True)
    assert [jsonable(r) for r in got2] == [jsonable(w) for w in want]
    final = many.engine.download()
    pk = many.packer
    ref = pk.pack_nodes(ora)
    for f in ("p0", "p1"):
        assert np.array_equal(getattr(final, f), getattr(ref, f)), f
    for f in ("gpu_free", "hp_free"):
        assert np.array_equal(final.p2[f], ref.p2[f]), f
    assert np.array_equal(final.p4["busy_time"], ref.p4["busy_time"])
