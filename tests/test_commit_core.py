"""Row f1 (commit step on packed state, nhd_amd/csrc/commit_core.h + seq_core.h), host build: against the fixtures the
unmodified reference produced (tests/golden/commit) and against the oracle's sequential batch.  CPU only."""
import copy
import os

import numpy as np
import pytest

from nhd_amd import pack
from workload import refmodel, synth
from oracle import nhd_oracle as O
from tests import commit_check, harness


@pytest.mark.parametrize("path", commit_check.FIXTURES, ids=[os.path.basename(p)[:-5] for p in commit_check.FIXTURES])
def test_host_twin_reproduces_reference_commits(path):
    case = commit_check.load(path)
    spec, nodes, tops, pk, table, reqs = commit_check.build(case)
    dict_before = pk.dict_version
    node, maps, places, status, done = harness.schedule(pk, table, reqs, case["clock"], apply=True)
    assert done == len(reqs)
    commit_check.check(case, nodes, tops, table, reqs, node, maps, places, status, table)
    commit_check.check_signatures(pk, table)
    assert pk.dict_version == dict_before        # the closure held every NIC state the commits produced


@pytest.mark.parametrize("cfg,n,P", [(3, 40, 120), (4, 16, 150), (5, 120, 300), (2, 20, 80)])
def test_placements_match_oracle_ids(cfg, n, P):
    """Physical ids and decisions of the host twin against the oracle's sequential batch (pinned to the reference in
    tests/test_mode_b_oracle.py) on more shapes than the committed fixtures cover."""
    spec = synth.make_cluster(cfg, n_nodes=n)
    pods, groups = synth.make_pods(cfg, n_pods=P)
    for p in pods:
        p["misc_smt"] = True
    nodes = spec.build_nodes()
    tops = [refmodel.make_topology(s) for s in pods]
    pk = pack.Packer()
    table = pk.pack_nodes(nodes)
    reqs = pk.digest_many(tops, groups)
    pk.close_signatures()
    node, maps, places, status, done = harness.schedule(pk, table, reqs, spec.clock_now, apply=True)
    assert done == P
    ids = []
    want = O.schedule_sequence(nodes, tops, groups, spec.clock_now, ids_out=ids)       # mutates `nodes`
    for i, (w, wid) in enumerate(zip(want, ids)):
        if w[0] is None:
            assert node[i] < 0
            continue
        assert table.names[int(node[i])] == w[0]
        nd = nodes[w[0]]
        G = int(reqs[i]["n_groups"])
        got = pack.expand_placement(places[i], G, int(nd.cores_per_proc), int(nd.cores_per_proc) * int(nd.sockets),
                                    [int(reqs[i]["gpus"][g]) for g in range(G)])
        assert got == wid, (i, got, wid)
    # the packed state after the batch == the packed form of the oracle's mutated nodes
    after = pack.Packer()
    t2 = after.pack_nodes(nodes)
    for f in ("p0", "p1"):
        assert np.array_equal(getattr(table, f), getattr(t2, f)), f
    for f in ("gpu_free", "hp_free"):
        assert np.array_equal(table.p2[f], t2.p2[f]), f
    assert np.array_equal(table.p4["busy_time"], t2.p4["busy_time"])
    assert pack.resolve_signatures(pk, table) == pack.resolve_signatures(after, t2)


def test_unknown_nic_state_is_reported_not_guessed():
    """Without the signature closure a commit can leave a node in a NIC state the dictionary has no id for: the batch
    stops right after that pod (n_done), the node's record says so, and interning + patching lets it continue."""
    spec = synth.make_cluster(3, n_nodes=6)
    pods, groups = synth.make_pods(3, n_pods=60)
    for p in pods:
        p["misc_smt"] = True
    nodes = spec.build_nodes()
    pk = pack.Packer()
    table = pk.pack_nodes(nodes)
    reqs = pk.digest_many([refmodel.make_topology(s) for s in pods], groups)
    node, maps, places, status, done = harness.schedule(pk, table, reqs, spec.clock_now, apply=True)
    if done < len(reqs):
        assert status[done - 1] == pack.COMMIT_NEW_SIG
        v = int(node[done - 1])
        before = pk.dict_version
        pk.sigs_from_detail(table.detail[v])
        assert pk.dict_version > before                      # it really was a new signature
    else:
        assert not (status == pack.COMMIT_NEW_SIG).any()


def test_single_commit_matches_oracle():
    """nhdfit_commit's arithmetic (one placement) on heterogeneous random nodes."""
    from tests import util
    nl = util.random_cluster(4242, 60)
    rng = np.random.default_rng(11)
    pk = pack.Packer()
    table = pk.pack_nodes(nl)
    pk.close_signatures()
    placed = 0
    for k in range(120):
        spec = util.random_pod_spec(rng, max_groups=3)
        spec["misc_smt"] = True
        top = refmodel.make_topology(spec)
        res = O.find_node(nl, top, util.CLOCK)
        if res[0] is None:
            continue
        i = table.names.index(res[0])
        req = pk.digest(top)
        m = np.zeros((), pack.MAPPING)
        G = len(res[1]["gpu"])
        m["gpu"][:G] = res[1]["gpu"]; m["cpu"][:G + 1] = res[1]["cpu"]
        m["nic_numa"][:G] = [x[0] for x in res[1]["nic"]]; m["nic_idx"][:G] = [x[1] for x in res[1]["nic"]]
        m["valid"] = 1
        ids = {}
        try:
            O.commit(nl[res[0]], top, res[1], util.CLOCK + k, ids)
        except O.CommitFailure:
            break                                            # parity undefined from here on (the reference unwinds, badly)
        rc, place = harness.commit(pk, table, i, req, m, util.CLOCK + k)
        assert rc in (pack.COMMIT_OK, pack.COMMIT_NEW_SIG)
        nd = nl[res[0]]
        got = pack.expand_placement(place, G, int(nd.cores_per_proc), int(nd.cores_per_proc) * int(nd.sockets),
                                    [int(req["gpus"][g]) for g in range(G)])
        assert got == ids, (k, got, ids)
        one = pack.empty_table(1)
        pack.Packer().pack_node_into(nd, one, 0)
        assert np.array_equal(one.p0[0], table.p0[i]) and np.array_equal(one.p1[0], table.p1[i])
        assert one.p2[0]["gpu_free"] == table.p2[i]["gpu_free"] and one.p2[0]["hp_free"] == table.p2[i]["hp_free"]
        placed += 1
    assert placed >= 20


@pytest.mark.parametrize("cfg,n,P", [(3, 40, 200), (4, 30, 250), (5, 80, 400)])
def test_misc_cores_without_the_smt_flag_take_second_threads(cfg, n, P):
    """The pods as the generator draws them (misc_cores_smt disabled for half): quirk Q1 lets one free physical core pass
    the filter for two misc cores, GetFreeCpuBatch's walk runs on into the sibling range (nhd/Node.py:502-519) and hands
    out the second thread as a core of its own - `misc_late` in the placement record.  Decisions, ids and the packed
    state after the batch against the oracle's loop; no pod may come back as 'would raise'."""
    spec = synth.make_cluster(cfg, n_nodes=n)
    pods, groups = synth.make_pods(cfg, n_pods=P)
    nodes = spec.build_nodes()
    tops = [refmodel.make_topology(s) for s in pods]
    pk = pack.Packer()
    table = pk.pack_nodes(nodes)
    reqs = pk.digest_many(tops, groups)
    pk.close_signatures()
    node, maps, places, status, done = harness.schedule(pk, table, reqs, spec.clock_now, apply=True)
    assert done == P and not (status == pack.COMMIT_WOULD_RAISE).any()
    ids = []
    want = O.schedule_sequence(nodes, tops, groups, spec.clock_now, ids_out=ids)
    late = 0
    for i, (w, wid) in enumerate(zip(want, ids)):
        if w[0] is None:
            assert node[i] < 0
            continue
        assert table.names[int(node[i])] == w[0]
        nd = nodes[w[0]]
        G = int(reqs[i]["n_groups"])
        got = pack.expand_placement(places[i], G, int(nd.cores_per_proc), int(nd.cores_per_proc) * int(nd.sockets),
                                    [int(reqs[i]["gpus"][g]) for g in range(G)])
        assert got == wid, (i, got, wid)
        late += int(places[i]["misc_late"] != 0)
    t2 = pack.Packer().pack_nodes(nodes)
    for f in ("p0", "p1"):
        assert np.array_equal(getattr(table, f), getattr(t2, f)), f


def test_run_on_walk_tiny_case():
    """The hand-made Q1 case of tests/test_seq_oracle.py (checked there against the unmodified reference: misc = [3, 11])."""
    from tests import util
    from tests.test_seq_oracle import _tiny_cluster
    pod = dict(map_type="NUMA", hugepages_gb=0, misc=2, misc_smt=False,
               groups=[dict(proc=2, helpers=0, rx=1, tx=1, gpus=[], proc_smt=False, helper_smt=False)])
    tops = [refmodel.make_topology(pod) for _ in range(2)]
    nodes = _tiny_cluster({})
    pk = pack.Packer()
    table = pk.pack_nodes(nodes)
    reqs = pk.digest_many(tops, None)
    pk.close_signatures()
    node, maps, places, status, done = harness.schedule(pk, table, reqs, util.CLOCK, apply=True)
    assert done == 2 and list(node) == [0, 1] and not status.any()
    got = pack.expand_placement(places[0], 1, 4, 8, [0])
    assert got == {"groups": [{"cores": [1, 2], "helpers": [], "gpus": []}], "misc": [3, 11]}
    assert int(places[0]["misc_late"]) == 1 << 3
    O.schedule_sequence(nodes, tops, [None, None], util.CLOCK)
    t2 = pack.Packer().pack_nodes(nodes)
    assert np.array_equal(table.p0, t2.p0) and np.array_equal(table.p1, t2.p1)


def test_claim_on_a_counter_outside_the_tracked_range_keeps_the_class():
    """pods_used = -5 on a free NIC (releases subtract one per pairing, claims add one per NIC): the packed
    counter says 'out of range', the reference's claim makes it -4 and the NIC keeps its capacity (nhd/Node.py:292)."""
    from tests import util
    nl = util.random_cluster(77, 12)
    name, nd = next((k, v) for k, v in nl.items() if len(v.nics) >= 1 and v.nics[0].numa_node < v.numa_nodes)
    nd.nics[0].pods_used = -5
    for c in nd.cores:
        c.used = c.core in getattr(nd, "reserved_cores", [])
    pk = pack.Packer()
    table = pk.pack_nodes(nl)
    pk.close_signatures()
    i = table.names.index(name)
    u, k = int(nd.nics[0].numa_node), int(nd.nics[0].idx)
    assert pack.get_pods(table.detail[i], u, k) == pack.PODS_LOST and table.detail[i]["nic_cls"][u][k] != 0
    top = refmodel.make_topology(dict(map_type="NUMA", hugepages_gb=0, misc=0, misc_smt=True,
                                      groups=[dict(proc=2, helpers=0, rx=1, tx=1, gpus=[], proc_smt=False, helper_smt=False)]))
    req = pk.digest(top)
    m = np.zeros((), pack.MAPPING)
    m["gpu"][0] = u; m["cpu"][:2] = (u, u); m["nic_numa"][0] = u; m["nic_idx"][0] = k; m["valid"] = 1
    rc, place = harness.commit(pk, table, i, req, m, util.CLOCK)
    assert rc == pack.COMMIT_OK
    O.commit(nd, top, {"gpu": (u,), "cpu": (u, u), "nic": [(u, k)]}, util.CLOCK)
    assert nd.nics[0].pods_used == -4
    one = pack.empty_table(1)
    fresh = pack.Packer()
    fresh.pack_node_into(nd, one, 0)
    assert fresh.caps[int(one.detail[0]["nic_cls"][u][k])] == pk.caps[int(table.detail[i]["nic_cls"][u][k])] != 0.0


def test_records_unpacked_in_one_pass_equal_the_field_by_field_reading():
    """pack.unpack_placements / unpack_mappings / unpack_big_mappings (one struct pass over the array) against the records read
    field by field, and expand_placement on both forms; expand_batch against the bit-by-bit walk it replaced (pair and late
    masks, bits up to 63)."""
    rng = np.random.default_rng(11)
    n = 64
    places = np.zeros(n, pack.PLACEMENT)
    raw = places.view(np.uint8).reshape(n, -1)
    raw[:] = rng.integers(0, 256, size=raw.shape, dtype=np.uint8)
    for f in ("proc_take", "proc_pair", "help_take", "help_pair", "proc_late", "help_late"):
        places[f] &= rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64) >> np.uint64(20)      # (sparse masks: a few cores per batch)
    places["numa"] = rng.integers(0, 2, size=(n, 5))
    rows = pack.unpack_placements(places)
    assert len(rows) == n
    for i in range(n):
        r = rows[i]
        assert list(r[0:4]) == places[i]["proc_take"].tolist() and list(r[4:8]) == places[i]["proc_pair"].tolist()
        assert list(r[8:12]) == places[i]["help_take"].tolist() and list(r[12:16]) == places[i]["help_pair"].tolist()
        assert r[16] == int(places[i]["misc_take"]) and r[17] == int(places[i]["misc_pair"])
        assert list(r[18:50]) == places[i]["gpu"].reshape(-1).tolist() and list(r[50:55]) == places[i]["numa"].tolist()
        assert r[55] == int(places[i]["status"])
        assert list(r[56:60]) == places[i]["proc_late"].tolist() and list(r[60:64]) == places[i]["help_late"].tolist() and r[64] == int(places[i]["misc_late"])
        G = int(rng.integers(1, 5))
        gp = [int(x) for x in rng.integers(0, 3, size=G)]
        assert pack.expand_placement(places[i], G, 64, 128, gp) == pack.expand_placement(r, G, 64, 128, gp)

    def walk(take, pair, numa, cpp, num_cores, late):            # the reference's order, bit by bit
        out = []
        for b in range(64):
            if take >> b & 1:
                out.append(numa * cpp + b)
                if pair >> b & 1:
                    out.append(numa * cpp + b + num_cores)
        for b in range(64):
            if late >> b & 1:
                out.append(numa * cpp + b + num_cores)
        return out
    for _ in range(200):
        take, pair, late = (int(x) for x in rng.integers(0, 2**63, size=3, dtype=np.uint64) & rng.integers(0, 2**63, size=3, dtype=np.uint64))
        take |= 1 << 63 if rng.random() < 0.1 else 0
        numa = int(rng.integers(0, 2))
        assert pack.expand_batch(take, pair, numa, 64, 128, late) == walk(take, pair, numa, 64, 128, late)

    maps = np.zeros(n, pack.MAPPING)
    maps.view(np.int8).reshape(n, -1)[:] = rng.integers(-1, 8, size=(n, pack.MAPPING.itemsize))
    for i, (gpu, cpu, nn, ni, valid) in enumerate(pack.unpack_mappings(maps)):
        assert list(gpu) == maps[i]["gpu"].tolist() and list(cpu) == maps[i]["cpu"].tolist() and list(nn) == maps[i]["nic_numa"].tolist()
        assert list(ni) == maps[i]["nic_idx"].tolist() and valid == int(maps[i]["valid"])
    big = np.zeros(n, pack.BIG_MAPPING)
    big.view(np.int8).reshape(n, -1)[:] = rng.integers(-1, 8, size=(n, pack.BIG_MAPPING.itemsize))
    for i, (gpu, cpu, nn, ni, valid) in enumerate(pack.unpack_big_mappings(big)):
        assert list(gpu) == big[i]["gpu"].tolist() and list(cpu) == big[i]["cpu"].tolist() and list(nn) == big[i]["nic_numa"].tolist()
        assert list(ni) == big[i]["nic_idx"].tolist() and valid == int(big[i]["valid"])
