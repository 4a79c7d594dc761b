"""Pins oracle/nhd_oracle.py (the CPU restatement) to the UNMODIFIED reference imported from
/root/reference.  Runs only where the reference tree exists (the build container)."""
import numpy as np
import pytest

from workload import refmodel, synth
from oracle import nhd_oracle as O
from tests import util


@pytest.fixture(scope="module")
def refclock(ref):
    from oracle import ref_loader
    return ref_loader.VirtualClock(util.CLOCK).install()


@pytest.mark.parametrize("seed", range(12))
def test_random_clusters_match_reference(ref, refclock, seed):
    from oracle import ref_loader
    nl = util.random_cluster(1000 + seed, 24, ref)
    rng = np.random.default_rng(seed)
    hits = 0
    for _ in range(25):
        spec = util.random_pod_spec(rng)
        top = refmodel.make_topology(spec, ref)
        want = ref_loader.find_node(nl, top)
        got = O.find_node(nl, top, util.CLOCK)
        assert got == want, (spec, got, want)
        hits += want[0] is not None
    assert hits > 0


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5])
def test_synthetic_configs_match_reference(ref, refclock, cfg):
    from oracle import ref_loader
    spec = synth.make_cluster(cfg, n_nodes=48)
    refclock.t = spec.clock_now
    nl = spec.build_nodes(ref)
    pods, pgroups = synth.make_pods(cfg, n_pods=12)
    for ps, pg in zip(pods, pgroups):
        top = refmodel.make_topology(ps, ref)
        sub = O.initial_node_filter(nl, pg)
        assert O.find_node(sub, top, spec.clock_now) == ref_loader.find_node(sub, top)
    refclock.t = util.CLOCK


def test_standins_parse_like_reference(ref):
    rng = np.random.default_rng(5)
    for i in range(40):
        lab = util.random_labels(rng)
        a = ref.Node("x"); assert a.ParseLabels(lab); a.SetHugepages(16, 9)
        b = refmodel.node_from_labels("x", lab, (16, 9))
        assert (a.sockets, a.numa_nodes, a.smt_enabled, a.cores_per_proc, a.groups, a.maintenance) == \
               (b.sockets, b.numa_nodes, b.smt_enabled, b.cores_per_proc, b.groups, b.maintenance)
        assert [(c.core, c.socket, c.sibling, c.used) for c in a.cores] == [(c.core, c.socket, c.sibling, c.used) for c in b.cores]
        assert [(n.ifname, n.speed, n.numa_node, n.pciesw, n.idx, n.mac) for n in a.nics] == \
               [(n.ifname, n.speed, n.numa_node, n.pciesw, n.idx, n.mac) for n in b.nics]
        assert [(g.device_id, g.numa_node, g.pciesw) for g in a.gpus] == [(g.device_id, g.numa_node, g.pciesw) for g in b.gpus]
        assert a.mem.free_hugepages_gb == b.mem.free_hugepages_gb


def test_oracle_same_on_standins_and_reference_objects(ref, refclock):
    """The oracle reads objects by attribute only: stand-ins and reference objects must agree."""
    rng = np.random.default_rng(77)
    nl_ref = util.random_cluster(4242, 20, ref)
    nl_std = util.random_cluster(4242, 20, None)
    for _ in range(30):
        spec = util.random_pod_spec(rng)
        assert O.find_node(nl_ref, refmodel.make_topology(spec, ref), util.CLOCK) == \
               O.find_node(nl_std, refmodel.make_topology(spec), util.CLOCK)
