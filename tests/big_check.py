"""Shared checker for tests/golden/big/*.json (oracle/gen_golden_big.py, the unmodified reference): pods with 5..8 processing
groups - more than the table-driven pass holds - mixed with ordinary pods, on ordinary nodes, on ordinary and wide nodes and on
nodes with interchangeable NICs, under both batch semantics: every pod against one snapshot (FindNode's answers; the per-node
verdicts through FindNode on one-node candidate sets) and the scheduler's loop (FindNode + commit, pod after pod: node, mapping,
the physical ids the reference wrote into the pod's topology, and every node's state afterwards)."""
import glob
import json
import os

from nhd_amd import pack
from tests import util
from tests.wide_check import as_jsonable, mirror_state
from workload import refmodel

FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "big", "*.json")))


def check(path, make_matcher, verdict_pods=6):
    with open(path) as f:
        case = json.load(f)
    nl = util.build_cluster(case["nodes"])
    tops = [refmodel.make_topology(s) for s in case["pods"]]
    big = [len(t.proc_groups) > pack.MAX_GROUPS for t in tops]
    assert sum(big) >= 5
    if "plain" in path or "mixed" in path:
        assert sum(big) < len(big)                                     # ordinary pods ride along in the same calls
    m = make_matcher(case["clock"])
    got = m.FindNodes(nl, tops)
    assert [as_jsonable(r) for r in got] == case["snapshot"]
    assert m.unmirrored == {}
    if "drawn_wide" in case:
        assert set(m.wide_nodes) <= set(case["drawn_wide"]) and len(m.wide_nodes) >= 3
    # per-node verdicts of the big pods: FindNode over one-node candidate sets of the attached mirror (a candidate mask per call)
    m.attach(nl)
    names = list(nl)
    checked = 0
    for p in [p for p in range(len(tops)) if big[p]][:verdict_pods]:
        for k, name in enumerate(names):
            res = m.FindNode({name: nl[name]}, tops[p])
            assert (res[0] is not None) == (case["feasible"][p][k] == "1"), (p, name, res)
            if res[0] is not None and case["snapshot"][p][0] == name:
                assert as_jsonable(res) == case["snapshot"][p]
        checked += 1
    assert checked >= min(verdict_pods, sum(big))
    # the scheduler's loop, decided and committed on the device
    seq = m.ScheduleBatch(nl, tops, now=case["clock"], apply=True)
    want = case["sequence"]
    assert [as_jsonable(r) for r in seq[:len(want)]] == [w[:2] for w in want]
    assert m.last_placements[:len(want)] == [w[2] if w[0] is not None else None for w in want]
    placed_big = sum(1 for w, b in zip(want, big) if b and w[0] is not None)
    assert placed_big >= 2, placed_big
    if len(want) == len(tops):
        state = mirror_state(m, nl, m.engine.download())
        for name, w in case["final"].items():
            assert state[name] == w, (name, state[name], w)
    return m
