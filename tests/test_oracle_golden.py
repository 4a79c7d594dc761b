"""The CPU restatement (oracle/nhd_oracle.py) against the committed golden vectors, which were
produced by the unmodified reference (oracle/gen_golden.py).  Runs anywhere (no reference, no GPU)."""
import glob
import json
import os

import pytest

from workload import refmodel
from oracle import nhd_oracle as O
from tests import util

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.json")))


def load_case(path):
    with open(path) as f:
        return json.load(f)


def as_tuple(res):
    if res[0] is None:
        return [None]
    m = res[1]
    return [res[0], {"gpu": list(m["gpu"]), "cpu": list(m["cpu"]), "nic": [list(x) for x in m["nic"]]}]


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-5] for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    case = load_case(path)
    nl = util.build_cluster(case["nodes"])            # stand-in objects: no reference needed
    for pod, want, feas in zip(case["pods"], case["expected"], case["feasible"]):
        top = refmodel.make_topology(pod["spec"])
        sub = O.initial_node_filter(nl, pod["groups"])
        assert as_tuple(O.find_node(sub, top, case["clock"])) == want
        got = "".join("1" if (n in sub and O.feasible(v, top, case["clock"])) else "0" for n, v in nl.items())
        assert got == feas


def test_golden_present():
    assert len(GOLDEN) >= 10
