"""The lone-pod form of the find (nhd_amd/csrc/fit_core.h: lone_pod_fits, lone_nic_bits, sig_reach_flat) on the host build:
one pod's verdict over every node from the pod's own 16-bit masks must equal the bit-sliced table form (hh_find, itself
pinned to the oracles) - full bitmaps, scores, and the NIC-feasible assignment bits the mapping of the winner starts from."""
import numpy as np
import pytest

from nhd_amd import pack
from tests import harness, util
from workload import planes, refmodel, synth


@pytest.mark.parametrize("cfg,n,P", [(1, 32, 8), (2, 1024, 96), (3, 2048, 160), (4, 3072, 200), (5, 2048, 200)], ids=["c1", "c2", "c3", "c4", "c5"])
def test_lone_form_equals_table_form_on_baseline_shapes(cfg, n, P):
    spec = synth.make_cluster(cfg, n_nodes=n)
    pods, groups = synth.make_pods(cfg, n_pods=P)
    tops = [refmodel.make_topology(s) for s in pods]
    pk = pack.Packer()
    table = planes.planes_from_spec(pk, spec)
    reqs = pk.digest_many(tops, groups)
    pk.close_signatures()                                                   # (more signatures than the mirror uses: every record is walked)
    rng = np.random.default_rng(cfg)
    for cand in (None, rng.integers(0, 2**63, size=(n + 63) // 64, dtype=np.uint64)):
        for now in (spec.clock_now, spec.clock_now + 31.0):
            s1, b1, _ = harness.find(pk, table, reqs, now, cand=cand, want_map=False)
            s2, b2, bits = harness.find_lone(pk, table, reqs, now, cand=cand)
            assert np.array_equal(b1, b2)
            assert np.array_equal(s1, s2)
            assert np.array_equal(bits != 0, s2 != 0)                        # a winner has at least one NIC-feasible assignment
    assert (s1 != 0).sum() > 0


@pytest.mark.parametrize("seed", range(8))
def test_lone_form_on_random_heterogeneous_clusters(seed):
    """Clusters wider than the BASELINE shapes (1- and 2-socket nodes, mixed NIC speeds, odd switch layouts, invalid and
    four-group pods among the requests)."""
    nl = util.random_cluster(300 + seed, 60)
    rng = np.random.default_rng(300 + seed)
    tops = [refmodel.make_topology(util.random_pod_spec(rng, max_groups=4 if seed % 2 == 0 else 3)) for _ in range(50)]
    pk = pack.Packer()
    table = pk.pack_nodes(nl)
    reqs = pk.digest_many(tops)
    s1, b1, _ = harness.find(pk, table, reqs, util.CLOCK, want_map=False)
    s2, b2, _ = harness.find_lone(pk, table, reqs, util.CLOCK)
    assert np.array_equal(b1, b2) and np.array_equal(s1, s2)
