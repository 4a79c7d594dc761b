"""Shared checker for tests/golden/sharing/*.json (oracle/gen_golden_sharing.py: the reference with nhd/Node.py:20 ENABLE_SHARING
flipped to True).  The module constant of the stand-in node objects is flipped the same way (workload.refmodel.ENABLE_SHARING);
the packer then mirrors every node for the general path with its NICs' speed_used (pack.WIDE_SHARE) and the path prices a NIC at
speed * 0.9 - speed_used[x].  Checked: every pod against one snapshot (FindNode's answers), the scheduler's loop decided and
committed on the mirror (node, mapping, physical ids), and every node afterwards - its NICs' speed_used bit for bit."""
import glob
import json
import os

from tests import util
from tests.wide_check import as_jsonable, mirror_state
from workload import refmodel

FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "sharing", "*.json")))


def check(path, make_matcher):
    with open(path) as f:
        case = json.load(f)
    saved = refmodel.ENABLE_SHARING
    refmodel.ENABLE_SHARING = True
    try:
        nl = util.build_cluster(case["nodes"])
        tops = [refmodel.make_topology(s) for s in case["pods"]]
        m = make_matcher(case["clock"])
        got = m.FindNodes(nl, tops)
        assert [as_jsonable(r) for r in got] == case["snapshot"]
        assert m.unmirrored == {} and len(m.wide_nodes) == len(nl)          # every node rides the general path
        m.attach(nl)
        seq = m.ScheduleBatch(nl, tops, now=case["clock"], apply=True)
        want = case["sequence"]
        assert [as_jsonable(r) for r in seq] == [w[:2] for w in want]
        assert m.last_placements == [w[2] if w[0] is not None else None for w in want]
        state = mirror_state(m, nl, m.engine.download())
        share = m.engine.wide_share_download()
        assert len(share) == len(nl)
        for i, (name, node) in enumerate(nl.items()):
            w = dict(case["final"][name])
            used = w.pop("speed_used")
            assert state[name] == w, (name, state[name], w)
            for nic, u in zip(node.nics, used):
                if 0 <= nic.numa_node < node.numa_nodes:
                    got_u = [float(share[i]["used"][nic.numa_node][nic.idx][x]) for x in range(2)]
                    assert got_u == u, (name, nic.numa_node, nic.idx, got_u, u)
        return m
    finally:
        refmodel.ENABLE_SHARING = saved
