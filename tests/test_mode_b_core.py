"""Mode B (sequential commit) resolver - host build of nhd_amd/csrc/seq_core.h - against the oracle's
sequential batch (itself pinned to the reference in tests/test_mode_b_oracle.py)."""
import numpy as np
import pytest

from nhd_amd import pack
from workload import refmodel, synth
from oracle import nhd_oracle as O
from tests import harness, util


def decode(names, reqs, node, maps):
    out = []
    for p in range(len(reqs)):
        if node[p] < 0:
            out.append((None,))
            continue
        G = int(reqs[p]["n_groups"])
        m = maps[p]
        out.append((names[node[p]], {"gpu": tuple(int(x) for x in m["gpu"][:G]), "cpu": tuple(int(x) for x in m["cpu"][:G + 1]),
                                     "nic": [(int(a), int(b)) for a, b in zip(m["nic_numa"][:G], m["nic_idx"][:G])]}))
    return out


def norm(res):
    return (None,) if res[0] is None else (res[0], {"gpu": tuple(res[1]["gpu"]), "cpu": tuple(res[1]["cpu"]),
                                                    "nic": [tuple(x) for x in res[1]["nic"]]})


@pytest.mark.parametrize("cfg,n,P", [(2, 40, 120), (3, 12, 120), (3, 40, 120), (4, 60, 200), (4, 16, 150), (5, 120, 200)])
def test_sequential_batch_matches_oracle(cfg, n, P):
    spec = synth.make_cluster(cfg, n_nodes=n)
    pods, groups = synth.make_pods(cfg, n_pods=P)
    for p in pods:
        p["misc_smt"] = True
    nl = spec.build_nodes()
    tops = [refmodel.make_topology(s) for s in pods]
    pk = pack.Packer()
    table = pk.pack_nodes(nl)
    reqs = pk.digest_many(tops, groups)
    node, maps, status = harness.resolve(pk, table, reqs, spec.clock_now)
    want = O.schedule_sequence(nl, tops, groups, spec.clock_now)        # mutates nl (after packing)
    assert decode(table.names, reqs, node, maps) == [norm(w) for w in want]
    assert not status.any()
    placed = sum(w[0] is not None for w in want)
    assert placed >= 10
    if n <= 16:
        assert placed < P            # the small clusters fill up: later pods are rejected or pushed elsewhere


@pytest.mark.parametrize("seed", range(6))
def test_sequential_batch_random_clusters(seed):
    nl = util.random_cluster(71000 + seed, 30, occupancy=0.15)
    rng = np.random.default_rng(seed)
    specs = []
    for _ in range(80):
        s = util.random_pod_spec(rng)
        s["misc_smt"] = True
        if s["map_type"] == "NONE":
            s["map_type"] = "NUMA"
        specs.append(s)
    tops = [refmodel.make_topology(s) for s in specs]
    pk = pack.Packer()
    table = pk.pack_nodes(nl)
    reqs = pk.digest_many(tops)
    node, maps, status = harness.resolve(pk, table, reqs, util.CLOCK)
    want = []
    t = util.CLOCK
    for top in tops:
        res = O.find_node(nl, top, t)
        ok = True
        if res[0] is not None:
            try:
                O.commit(nl[res[0]], top, res[1], t)
            except O.CommitFailure:
                ok = False
        want.append((norm(res), ok))
        if not ok:
            break                      # parity is undefined once the reference's commit step would have raised
    k = len(want)
    got = decode(table.names, reqs, node, maps)
    assert got[:k - (0 if want[-1][1] else 1)] == [w[0] for w in want][:k - (0 if want[-1][1] else 1)]
