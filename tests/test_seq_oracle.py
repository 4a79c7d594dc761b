"""Pins oracle/seq_oracle.py (mode B at BASELINE sizes: C scan + C commit + Python mapping of the winner) to
  * oracle/nhd_oracle.schedule_sequence (itself pinned to the unmodified reference, tests/test_mode_b_oracle.py):
    decisions, mappings, physical ids and the final state of every node;
  * the reference-generated commit fixtures tests/golden/commit/*.json (decisions, mappings, physical ids);
  * the unmodified reference's own loop where /root/reference is present (build container)."""
import contextlib
import io
import os

import numpy as np
import pytest

from oracle import coracle, seq_oracle
from oracle import nhd_oracle as O
from tests import commit_check, util
from workload import refmodel, synth


def flat_state(sc, i):
    nd = sc.nodes[i]
    co, nc, go, ng, no, nn = (int(nd[k]) for k in ("core_off", "n_cores", "gpu_off", "n_gpus", "nic_off", "n_nics"))
    return ([bool(x) for x in sc.a["core_used"][co:co + nc]], [bool(x) for x in sc.a["gpu_used"][go:go + ng]],
            [int(x) for x in sc.a["nic_pods"][no:no + nn]], int(nd["hp_free"]), float(nd["busy_time"]))


def object_state(n):
    return ([bool(c.used) for c in n.cores], [bool(g.used) for g in n.gpus], [int(k.pods_used) for k in n.nics],
            int(n.mem.free_hugepages_gb), float(n.busy_time))


def norm_map(m):
    return None if m is None else {"gpu": [int(x) for x in m["gpu"]], "cpu": [int(x) for x in m["cpu"]], "nic": [[int(a), int(b)] for a, b in m["nic"]]}


@pytest.mark.parametrize("cfg,n,P,safe", [(2, 24, 80, True), (3, 40, 150, True), (4, 60, 200, True), (5, 80, 250, True)])
def test_against_python_oracle_synth(cfg, n, P, safe):
    spec = synth.make_cluster(cfg, n_nodes=n)
    pods, groups = synth.make_pods(cfg, n_pods=P)
    if safe:
        for p in pods:
            p["misc_smt"] = True
    tops = [refmodel.make_topology(p) for p in pods]
    nl = spec.build_nodes()
    names = list(nl)
    sc = seq_oracle.SeqCluster(coracle.Cluster.from_nodes(nl), names)
    win, maps, ids, n_def = seq_oracle.schedule_sequence(sc, tops, groups, spec.clock_now)
    want_ids = []
    try:
        want = O.schedule_sequence(nl, tops[:n_def], groups[:n_def], spec.clock_now, ids_out=want_ids)
    except O.CommitFailure:
        pytest.fail("the Python oracle raises inside the prefix the C commit called defined")
    assert [None if w < 0 else names[w] for w in win] == [r[0] for r in want]
    assert [norm_map(m) for m in maps] == [norm_map(r[1]) if r[0] is not None else None for r in want]
    assert ids == want_ids
    if n_def < P:                                        # the next pod is where the reference raises
        nl2 = spec.build_nodes()
        with pytest.raises(O.CommitFailure):
            O.schedule_sequence(nl2, tops[:n_def + 1], groups[:n_def + 1], spec.clock_now)
    else:
        for i, name in enumerate(names):
            assert flat_state(sc, i) == object_state(nl[name]), name
    assert sum(w >= 0 for w in win) >= 10
    assert n_def == P


def _tiny_cluster(extra_labels, k=3):
    lab = {refmodel.NFD + "nfd-extras-cpu.numSockets": "2", refmodel.NFD + "nfd-extras-cpu.num_cores": "8",
           refmodel.NFD + "cpu-hardware_multithreading": "true",
           refmodel.NFD + "nfd-extras-nic.eth0.mlx.0000000000aa.100000Mbs.0.10.0.0": "true",
           refmodel.NFD + "nfd-extras-nic.eth1.mlx.0000000000bb.100000Mbs.1.20.1.0": "true",
           "DATA_PLANE_VLAN": "1", "DATA_DEFAULT_GW": "10.0.0.1/32"}
    lab.update(extra_labels)
    nl = {}
    for i in range(k):
        n = refmodel.node_from_labels(f"n{i}", lab, (64, 64))
        n.cores[0].used = n.cores[8].used = True          # socket 0: 3 free physical cores (1, 2, 3)
        for c in (4, 5, 6, 7, 12, 13, 14, 15):
            n.cores[c].used = True                        # socket 1: none
        nl[n.name] = n
    return nl


def test_misc_cores_run_on_into_the_sibling_range():
    """Quirk Q1: on an SMT node the filter halves the pod-level misc cores whatever their SMT flag says (Matcher.py:198), the
    commit uses the real flag (Node.py:799).  One free physical core passes the filter for two non-SMT misc cores;
    GetFreeCpuBatch marks nothing while it scans (Node.py:502-519), so its walk over Node.cores runs on into the sibling
    range and hands out the free core's second thread as a core of its own: [3, 11] - no exception, defined behaviour."""
    pod = dict(map_type="NUMA", hugepages_gb=0, misc=2, misc_smt=False,
               groups=[dict(proc=2, helpers=0, rx=1, tx=1, gpus=[], proc_smt=False, helper_smt=False)])
    tops = [refmodel.make_topology(pod) for _ in range(2)]
    nl = _tiny_cluster({})
    sc = seq_oracle.SeqCluster(coracle.Cluster.from_nodes(nl), list(nl))
    win, maps, ids, n_def = seq_oracle.schedule_sequence(sc, tops, [None] * 2, util.CLOCK)
    assert n_def == 2 and win == [0, 1]
    assert ids[0] == {"groups": [{"cores": [1, 2], "helpers": [], "gpus": []}], "misc": [3, 11]}
    want_ids = []
    want = O.schedule_sequence(_tiny_cluster({}), tops, [None] * 2, util.CLOCK, ids_out=want_ids)
    assert [r[0] for r in want] == ["n0", "n1"] and want_ids == ids


def test_stops_where_the_reference_raises():
    """Quirk Q2 counts GROUPS per PCIe switch (Matcher.py:312-322), the commit takes every GPU of a PCI-mode group from
    its NIC's switch (Node.py:648-655, 703-712): a group with two GPUs passes the filter on a switch with one free GPU and
    SetPhysicalIdsFromMapping raises IndexError.  The sequence is defined up to that pod."""
    gpus = {refmodel.NFD + "nfd-extras-gpu.0.V100.0.10": "true", refmodel.NFD + "nfd-extras-gpu.1.V100.0.11": "true"}
    ok = dict(map_type="PCI", hugepages_gb=0, misc=0, misc_smt=True,
              groups=[dict(proc=2, helpers=0, rx=1, tx=1, gpus=[0], proc_smt=True, helper_smt=False)])
    bad = dict(map_type="PCI", hugepages_gb=0, misc=0, misc_smt=True,
               groups=[dict(proc=2, helpers=0, rx=1, tx=1, gpus=[0, 0], proc_smt=True, helper_smt=False)])
    tops = [refmodel.make_topology(p) for p in (ok, bad, ok)]
    nl = _tiny_cluster(gpus)
    sc = seq_oracle.SeqCluster(coracle.Cluster.from_nodes(nl), list(nl))
    win, maps, ids, n_def = seq_oracle.schedule_sequence(sc, tops, [None] * 3, util.CLOCK)
    assert n_def == 1 and win == [0]
    with pytest.raises(O.CommitFailure):
        O.schedule_sequence(_tiny_cluster(gpus), tops, [None] * 3, util.CLOCK)


@pytest.mark.parametrize("seed", range(6))
def test_against_python_oracle_heterogeneous(seed):
    """Random clusters wider than the BASELINE configs (1- and 2-socket nodes, mixed NIC speeds, odd switch layouts)."""
    rng = np.random.default_rng(900 + seed)
    nl = util.random_cluster(900 + seed, 40)
    pods = [util.random_pod_spec(rng) for _ in range(120)]
    for p in pods:
        p["misc_smt"] = True
    tops = [refmodel.make_topology(p) for p in pods]
    names = list(nl)
    sc = seq_oracle.SeqCluster(coracle.Cluster.from_nodes(nl), names)
    win, maps, ids, n_def = seq_oracle.schedule_sequence(sc, tops, [None] * len(tops), util.CLOCK)
    want_ids = []
    try:
        want = O.schedule_sequence(nl, tops[:n_def], [None] * n_def, util.CLOCK, ids_out=want_ids)
    except (O.CommitFailure, IndexError):
        pytest.skip("this seed runs into a commit the reference raises on")
    assert [None if w < 0 else names[w] for w in win] == [r[0] for r in want]
    assert [norm_map(m) for m in maps] == [norm_map(r[1]) if r[0] is not None else None for r in want]
    assert ids == want_ids


@pytest.mark.parametrize("path", commit_check.FIXTURES, ids=[os.path.basename(p)[:-5] for p in commit_check.FIXTURES])
def test_against_reference_commit_fixtures(path):
    """tests/golden/commit: what the unmodified reference decided, mapped and wrote into the pods' topologies."""
    case = commit_check.load(path)
    spec = synth.make_cluster(case["config"], n_nodes=case["n_nodes"])
    pods, groups = synth.make_pods(case["config"], n_pods=case.get("n_pods_drawn", case["n_pods"]))
    pods, groups = pods[:case["n_pods"]], groups[:case["n_pods"]]
    if case.get("force_misc_smt", True):                       # (commit_q1_*: the pods as drawn - quirk Q1's run-on walk)
        for p in pods:
            p["misc_smt"] = True
    tops = [refmodel.make_topology(p) for p in pods]
    sc = seq_oracle.SeqCluster(coracle.Cluster.from_spec(spec), [spec.name(i) for i in range(spec.n)])
    win, maps, ids, n_def = seq_oracle.schedule_sequence(sc, tops, groups, case["clock"])
    assert n_def == len(tops)
    for i, want in enumerate(case["expected"]):
        if want[0] is None:
            assert win[i] < 0, i
            continue
        assert sc.name(win[i]) == want[0], (i, win[i], want[0])
        assert norm_map(maps[i]) == want[1], i
        assert ids[i] == want[2], i
    # final state: free-core masks, GPU mask, hugepages, busy time, claimed NICs as the reference left them
    for k in range(spec.n):
        want = case["final"][spec.name(k)]
        nd = sc.nodes[k]
        co, phys = int(nd["core_off"]), int(nd["n_scan"])
        cpp = phys // 2
        used = sc.a["core_used"]
        t0 = [sum((not used[co + s * cpp + b]) << b for b in range(cpp)) for s in range(2)]
        assert t0 == want["t0"], (k, t0, want["t0"])
        if nd["smt"]:
            t1 = [sum((not used[co + phys + s * cpp + b]) << b for b in range(cpp)) for s in range(2)]
            assert t1 == want["t1"], k
        go, ng = int(nd["gpu_off"]), int(nd["n_gpus"])
        assert sum((not sc.a["gpu_used"][go + g]) << g for g in range(ng)) == want["gpu_free"], k
        assert int(nd["hp_free"]) == want["hp_free"] and float(nd["busy_time"]) == want["busy_time"], k
        no, K = int(nd["nic_off"]), spec.nics_per_numa
        claimed = [[bool(sc.a["nic_pods"][no + u * K + j] > 0) for j in range(K)] for u in range(2)]
        assert claimed == want["nic_claimed"], k


def test_against_unmodified_reference(ref):
    """The reference's own loop (FindNode -> SetBusy -> SetPhysicalIdsFromMapping -> ClaimPodNICResources) on its own objects."""
    from oracle import ref_loader
    clock = ref_loader.VirtualClock(1.0e6).install()
    spec = synth.make_cluster(4, n_nodes=120)
    pods, groups = synth.make_pods(4, n_pods=260)
    for p in pods:
        p["misc_smt"] = True
    ref_nodes = spec.build_nodes(ref)
    tops_r = [refmodel.make_topology(p, ref) for p in pods]
    tops_o = [refmodel.make_topology(p) for p in pods]
    want = []
    for top, grp in zip(tops_r, groups):
        res = ref_loader.find_node(O.initial_node_filter(ref_nodes, grp), top)
        want.append(res)
        if res[0] is not None:
            n = ref_nodes[res[0]]
            n.SetBusy()
            with contextlib.redirect_stdout(io.StringIO()):
                nic_list = n.SetPhysicalIdsFromMapping(res[1], top)
            n.ClaimPodNICResources(list({x[0] for x in nic_list}))
    names = [spec.name(i) for i in range(spec.n)]
    sc = seq_oracle.SeqCluster(coracle.Cluster.from_spec(spec), names)
    win, maps, ids, n_def = seq_oracle.schedule_sequence(sc, tops_o, groups, clock.t)
    assert n_def == len(pods)
    assert [None if w < 0 else names[w] for w in win] == [r[0] for r in want]
    assert [norm_map(m) for m in maps] == [norm_map(r[1]) if r[0] is not None else None for r in want]
    for i, name in enumerate(names):
        n = ref_nodes[name]
        assert flat_state(sc, i) == ([bool(c.used) for c in n.cores], [bool(g.used) for g in n.gpus], [int(k.pods_used) for k in n.nics],
                                     int(n.mem.free_hugepages_gb), float(n.busy_time)), name
    assert sum(w >= 0 for w in win) >= 100


def test_sibling_range_walk_in_the_unmodified_reference(ref):
    """The Q1 case of test_misc_cores_run_on_into_the_sibling_range on the reference's own Node / CfgTopology objects."""
    from oracle import ref_loader
    ref_loader.VirtualClock(util.CLOCK).install()
    pod = dict(map_type="NUMA", hugepages_gb=0, misc=2, misc_smt=False,
               groups=[dict(proc=2, helpers=0, rx=1, tx=1, gpus=[], proc_smt=False, helper_smt=False)])
    proto = _tiny_cluster({}, k=1)["n0"]
    n = ref.Node("n0", True)
    lab = {refmodel.NFD + "nfd-extras-cpu.numSockets": "2", refmodel.NFD + "nfd-extras-cpu.num_cores": "8",
           refmodel.NFD + "cpu-hardware_multithreading": "true",
           refmodel.NFD + "nfd-extras-nic.eth0.mlx.0000000000aa.100000Mbs.0.10.0.0": "true",
           refmodel.NFD + "nfd-extras-nic.eth1.mlx.0000000000bb.100000Mbs.1.20.1.0": "true",
           "DATA_PLANE_VLAN": "1", "DATA_DEFAULT_GW": "10.0.0.1/32"}
    assert n.ParseLabels(lab)
    n.SetHugepages(64, 64)
    for c, pc in zip(n.cores, proto.cores):
        c.used = pc.used
    top = refmodel.make_topology(pod, ref)
    res = ref_loader.find_node({"n0": n}, top)
    assert res[0] == "n0"
    n.SetBusy()
    with contextlib.redirect_stdout(io.StringIO()):
        n.SetPhysicalIdsFromMapping(res[1], top)
    assert [c.core for c in top.proc_groups[0].proc_cores] == [1, 2]
    assert [c.core for c in top.misc_cores] == [3, 11]
