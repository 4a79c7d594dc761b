"""oracle/nhd_oracle.c (plain-C restatement) against oracle/nhd_oracle.py and the golden vectors."""
import glob
import json
import os
import time

import numpy as np
import pytest

from workload import refmodel, synth
from oracle import coracle
from oracle import nhd_oracle as O
from tests import util

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.json")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-5] for p in GOLDEN])
def test_c_oracle_reproduces_golden(path):
    with open(path) as f:
        case = json.load(f)
    nl = util.build_cluster(case["nodes"])
    cl = coracle.Cluster.from_nodes(nl)
    tops = [refmodel.make_topology(p["spec"]) for p in case["pods"]]
    pods = cl.pods_from_tops(tops, [p["groups"] for p in case["pods"]])
    winner, feas = cl.find(pods, case["clock"])
    names = list(nl)
    assert ["".join(str(int(x)) for x in row) for row in feas] == case["feasible"]
    assert [names[w] if w >= 0 else None for w in winner] == [e[0] for e in case["expected"]]


@pytest.mark.parametrize("seed", range(8))
def test_c_oracle_matches_python_oracle(seed):
    nl = util.random_cluster(52000 + seed, 30)
    rng = np.random.default_rng(seed)
    tops = [refmodel.make_topology(util.random_pod_spec(rng, max_groups=4)) for _ in range(30)]
    cl = coracle.Cluster.from_nodes(nl)
    winner, feas = cl.find(cl.pods_from_tops(tops), util.CLOCK)
    names = list(nl)
    for p, top in enumerate(tops):
        want = O.find_node(nl, top, util.CLOCK)
        assert (names[winner[p]] if winner[p] >= 0 else None) == want[0]
        assert list(feas[p]) == [int(O.feasible(v, top, util.CLOCK)) for v in nl.values()]


@pytest.mark.parametrize("cfg", [2, 3, 4, 5])
def test_spec_flattening_equals_object_flattening(cfg):
    spec = synth.make_cluster(cfg, n_nodes=150)
    a = coracle.Cluster.from_nodes(spec.build_nodes())
    b = coracle.Cluster.from_spec(spec)
    pods, groups = synth.make_pods(cfg, n_pods=40)
    tops = [refmodel.make_topology(s) for s in pods]
    wa, fa = a.find(a.pods_from_tops(tops, groups), spec.clock_now)
    wb, fb = b.find(b.pods_from_tops(tops, groups), spec.clock_now)
    assert np.array_equal(wa, wb) and np.array_equal(fa, fb)
    assert fa.sum() > 0
