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
