"""Shared checker for tests/golden/beyond/beyond_layout.json (oracle/gen_golden.py, the unmodified reference): a cluster with a
four-socket node and a node of 96 physical cores per socket among ordinary ones.  The fast layout (two sockets of 64 cores) cannot
hold those two; the general path (nhd_amd/csrc/wide_core.h) does: the product answers for the WHOLE cluster exactly as the reference
does - the 19 placements the reference makes on the two big nodes included - and nothing is left unmirrored."""
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
    want_all = case["expected_whole"]
    assert [as_jsonable(r) for r in got] == want_all
    assert m.unmirrored == {} and sorted(m.wide_nodes) == sorted(case["unmirrored"])     # (the fixture's key: the two nodes beyond the fast layout)
    on_big = sum(1 for w in want_all if w[0] in case["unmirrored"])
    assert on_big >= 10 and sum(a != b for a, b in zip(case["expected"], want_all)) > 0       # the reference does use the two nodes
    for top, grp, want in zip(tops, groups, want_all):            # the drop-in form: InitialNodeFilter, then FindNode per pod
        assert as_jsonable(m.FindNode(O.initial_node_filter(nl, grp), top)) == want
    m.attach(nl)                                                  # and with the persistent mirror
    for top, grp, want in zip(tops, groups, want_all):
        assert as_jsonable(m.FindNodes(nl, [top], pod_groups=[grp])[0]) == want
        assert as_jsonable(m.FindNode(O.initial_node_filter(nl, grp), top)) == want
    reqs = m.packer.digest_many(tops, groups)
    score, bm, _ = m.engine.find(reqs, case["clock"], want_bitmap=True, want_map=False)
    rows = unpack_bitmap(bm, len(nl))
    for i, row in enumerate(rows):                                # the verdict matrix: the reference's verdict for every (pod, node), the two included
        assert "".join(str(int(x)) for x in row) == case["feasible"][i], i
    return m
