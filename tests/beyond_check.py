"""Shared checker for tests/golden/beyond/beyond_layout.json (oracle/gen_golden.py, the unmodified reference): a cluster with a
four-socket node and a node of 96 physical cores per socket among ordinary ones.  The product cannot mirror those two; it must
answer for every other node exactly as the reference does for them, name the two, and never raise."""
import json
import os

import numpy as np

from oracle import nhd_oracle as O
from tests import util
from workload import refmodel

FIXTURE = os.path.join(os.path.dirname(__file__), "golden", "beyond", "beyond_layout.json")


def as_jsonable(res):
    if res[0] is None:
        return [None]
    m = res[1]
    return [res[0], {"gpu": list(m["gpu"]), "cpu": list(m["cpu"]), "nic": [list(x) for x in m["nic"]]}]


def check(make_matcher, unpack_bitmap):
    with open(FIXTURE) as f:
        case = json.load(f)
    nl = util.build_cluster(case["nodes"])
    tops = [refmodel.make_topology(p["spec"]) for p in case["pods"]]
    groups = [p["groups"] for p in case["pods"]]
    m = make_matcher(case["clock"])
    got = m.FindNodes(nl, tops, pod_groups=groups)
    assert [as_jsonable(r) for r in got] == case["expected"]
    assert sorted(m.unmirrored) == sorted(case["unmirrored"])
    differ = sum(a != b for a, b in zip(case["expected"], case["expected_whole"]))
    assert differ > 0                                             # (the reference itself does use the two nodes: the degradation is real)
    for top, grp, want in zip(tops, groups, case["expected"]):    # the drop-in form: InitialNodeFilter, then FindNode per pod
        assert as_jsonable(m.FindNode(O.initial_node_filter(nl, grp), top)) == want
    m.attach(nl)                                                  # and with the persistent mirror
    for top, grp, want in zip(tops, groups, case["expected"]):
        assert as_jsonable(m.FindNodes(nl, [top], pod_groups=[grp])[0]) == want
        assert as_jsonable(m.FindNode(O.initial_node_filter(nl, grp), top)) == want
    reqs = m.packer.digest_many(tops, groups)
    score, bm, _ = m.engine.find(reqs, case["clock"], want_bitmap=True, want_map=False)
    rows = unpack_bitmap(bm, len(nl))
    names = list(nl)
    for i, row in enumerate(rows):
        want = "".join("0" if names[j] in case["unmirrored"] else case["feasible"][i][j] for j in range(len(names)))
        assert "".join(str(int(x)) for x in row) == want, i
    return m
