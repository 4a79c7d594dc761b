"""Pods with 5..8 processing groups straight against the UNMODIFIED reference imported from /root/reference (build container
only): the oracle's answer (which the CPU suite holds the product to everywhere else) and the product's own - HipMatcher on the host
build of the general path (nhd_amd/csrc/wide_core.h over nhdfit_big_req) - must be the reference Matcher's `(name, mapping)` /
`(None,)` on random clusters of ordinary and wide nodes.  NIC counts are kept small: the reference makes (sum of NICs)^G deepcopies
per pod and node (nhd/Matcher.py:254)."""
import numpy as np
import pytest

from nhd_amd.matcher import HipMatcher
from oracle import nhd_oracle as O
from tests import harness, util
from tests.test_big_core import big_spec
from tests.test_wide_core import norm
from workload import refmodel


@pytest.fixture(scope="module")
def refclock(ref):
    from oracle import ref_loader
    return ref_loader.VirtualClock(util.CLOCK).install()


def few_nics(descs, most):
    for d in descs:
        keep, lab = 0, {}
        for k, v in d["labels"].items():
            if "nfd-extras-nic" in k:
                keep += 1
                if keep > most:
                    continue
            lab[k] = v
        d["labels"] = lab
        d["nic_pods_used"] = d["nic_pods_used"][:sum(1 for k in lab if "nfd-extras-nic" in k and "10000Mbs" not in k.replace("100000Mbs", ""))]
    return descs


@pytest.mark.parametrize("seed", range(6))
def test_big_pods_match_the_reference(ref, refclock, seed):
    from oracle import ref_loader
    wide = 0.35 if seed % 2 else 0.0
    descs = few_nics(util.mixed_cluster_desc(5100 + seed, 14, wide_share=wide, occupancy=0.08), 3)
    if wide:                                             # (four-socket nodes: six groups at most - the reference's own list scans grow as 4^G x 4^(G+1))
        hi = 6
    else:
        hi = 7
    nl_ref = util.build_cluster(descs, ref)
    nl = util.build_cluster(descs)
    rng = np.random.default_rng(900 + seed)
    specs = [big_spec(rng, 5, hi) for _ in range(7)]
    m = HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine)
    got = m.FindNodes(nl, [refmodel.make_topology(s) for s in specs])
    placed = 0
    for s, g in zip(specs, got):
        want = ref_loader.find_node(nl_ref, refmodel.make_topology(s, ref))
        assert O.find_node(nl, refmodel.make_topology(s), util.CLOCK) == want, s
        assert norm(g) == norm(want), s
        placed += want[0] is not None
    assert placed >= 1
