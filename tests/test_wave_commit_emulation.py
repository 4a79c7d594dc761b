"""The WAVEFRONT form of the commit step - the device source text of nhd_amd/csrc/seq2_kernel.h (commit_node_wave and its helpers:
what k_decide's speculators and workers run) cut out of the kernel header and executed on the host by 64 threads emulating the
lanes (tests/harness/wave_emul.cpp) - against the scalar form (commit_core.h commit_node: the host twin's and k_commit's): node
state, detail record, placement record and status byte for byte, on placements found for random pods on random clusters and on
the same placement committed again and again until the node runs dry (the would-raise statuses)."""
import copy

import numpy as np
import pytest

from nhd_amd import pack
from tests import harness, util
from workload import refmodel, synth


def _rows(table, i):
    return [bytes(np.ascontiguousarray(getattr(table, f)[i:i + 1]).tobytes()) for f in ("p0", "p1", "p2", "p3", "p4", "detail")]


def _both(pk, table, i, req, mp, bt):
    ta, tb = copy.deepcopy(table), copy.deepcopy(table)
    sa, pa = harness.commit(pk, ta, i, req, mp, bt)
    sb, pb = harness.wave_commit(pk, tb, i, req, mp, bt)
    assert sb != -100, "the lanes of the wavefront form disagree on the status"
    assert sa == sb, (sa, sb)
    assert pa.tobytes() == pb.tobytes(), (pa, pb)
    assert _rows(ta, i) == _rows(tb, i)
    tc = copy.deepcopy(table)                                  # the two-stage form (commit_summary_wave + commit_picks_wave; pods without GPUs)
    sc, pc = harness.wave_commit(pk, tc, i, req, mp, bt, form=2)
    assert sc == sa and pc.tobytes() == pa.tobytes() and _rows(tc, i) == _rows(ta, i), ("two-stage", sa, sc, pa, pc)
    return sa, ta


def _run(nl, specs, repeats, seed):
    tops = [refmodel.make_topology(s) for s in specs]
    pk = pack.Packer()
    table = pk.pack_nodes(nl)
    reqs = pk.digest_many(tops)
    pk.close_signatures()
    score, _, maps = harness.find(pk, table, reqs, util.CLOCK, want_bitmap=False)
    checked = raised = 0
    for p in np.flatnonzero(score != 0):
        i = int(0x7FFFFFFFFFFFFFFF - (int(score[p]) & 0x7FFFFFFFFFFFFFFF))
        if table.wide and i in table.wide:
            continue
        t = table
        for _ in range(repeats):                              # the same placement again: sooner or later cores / GPUs / NICs run out
            st, t = _both(pk, t, i, reqs[p], maps[p], util.CLOCK + 1.0)
            checked += 1
            raised += st == pack.COMMIT_WOULD_RAISE
    return checked, raised


@pytest.mark.parametrize("seed", range(4))
def test_wavefront_commit_equals_the_scalar_commit_on_random_clusters(seed):
    rng = np.random.default_rng(4100 + seed)
    nl = util.random_cluster(61000 + seed, 24, occupancy=0.15)
    specs = []
    for _ in range(24):
        s = util.random_pod_spec(rng, max_groups=4 if seed % 2 else 3)
        if s["map_type"] == "NONE":
            s["map_type"] = "PCI"
        specs.append(s)
    checked, raised = _run(nl, specs, repeats=4, seed=seed)
    assert checked >= 20 and raised >= 1


def test_wavefront_commit_on_the_baseline_mixes():
    """config 4 (GPUs, NICs, PCI locality) and config 5 (eight VFs per NUMA node: the signature keys over many NICs)"""
    total = 0
    for cfg in (4, 5):
        spec = synth.make_cluster(cfg, n_nodes=48)
        nl = spec.build_nodes()
        pods, _ = synth.make_pods(cfg, n_pods=28)
        tops = [refmodel.make_topology(s) for s in pods]
        pk = pack.Packer()
        table = pk.pack_nodes(nl)
        reqs = pk.digest_many(tops)
        pk.close_signatures()
        score, _, maps = harness.find(pk, table, reqs, spec.clock_now, want_bitmap=False)
        for p in np.flatnonzero(score != 0):
            i = int(0x7FFFFFFFFFFFFFFF - (int(score[p]) & 0x7FFFFFFFFFFFFFFF))
            t = table
            for _ in range(3):
                _, t = _both(pk, t, i, reqs[p], maps[p], spec.clock_now)
                total += 1
    assert total >= 40


@pytest.mark.parametrize("seed", range(3))
def test_wavefront_mapping_equals_the_scalar_mapping(seed):
    """seq_kernel.h map_on_state_wave (what k_seq's wavefronts and k_decide's speculators verify a candidate with: the tuple codes, NIC
    choices and table rows spread over the lanes) against seq_core.h map_on_state, for every (pod, node) pair of a random cluster - the
    feasible and the infeasible ones - with the set model alone, with the ascending-set table and with every table the device uses."""
    rng = np.random.default_rng(5200 + seed)
    nl = util.random_cluster(62000 + seed, 14, occupancy=float(rng.choice([0.0, 0.2, 0.4])))
    specs = []
    for _ in range(12):
        s = util.random_pod_spec(rng, max_groups=4 if seed % 2 else 3)
        if s["map_type"] == "NONE":
            s["map_type"] = "PCI"
        specs.append(s)
    tops = [refmodel.make_topology(s) for s in specs]
    pk = pack.Packer()
    table = pk.pack_nodes(nl)
    reqs = pk.digest_many(tops)
    pk.close_signatures()
    pairs = mapped = 0
    for i in range(table.n):
        if table.wide and i in table.wide:
            continue
        for p in range(len(reqs)):
            for tables in (0, 1, 2):
                rc, ok, ms, mw = harness.wave_map_on_state(pk, table, i, reqs[p], tables)
                assert rc == 0, (rc, ok, i, specs[p], ms, mw, tables)
            pairs += 1
            mapped += ok == 3
    assert pairs >= 100 and mapped >= 5


@pytest.mark.parametrize("seed", range(3))
def test_lone_pod_winner_mapped_by_the_wavefront_form(seed):
    """What a wave-cooperative mapping tail of the one-pod launch (k_find1) would compute: the winner of the lone-pod form, its
    NIC-feasible assignments from the pod's own masks (fit_core.h lone_nic_bits), map_on_state_wave on the winner's state - against
    the mapping the table pass returns for the same pod (the mode-A roles' answer, pinned to the reference elsewhere)."""
    rng = np.random.default_rng(5300 + seed)
    nl = util.random_cluster(63000 + seed, 30, occupancy=0.2)
    specs = [util.random_pod_spec(rng, max_groups=3) for _ in range(30)]
    tops = [refmodel.make_topology(s) for s in specs]
    pk = pack.Packer()
    table = pk.pack_nodes(nl)
    reqs = pk.digest_many(tops)
    score, _, maps = harness.find(pk, table, reqs, util.CLOCK, want_bitmap=False)
    ls, _, bits = harness.find_lone(pk, table, reqs, util.CLOCK)
    assert np.array_equal(ls, score)
    mapped = 0
    for p in np.flatnonzero(score != 0):
        i = int(0x7FFFFFFFFFFFFFFF - (int(score[p]) & 0x7FFFFFFFFFFFFFFF))
        if table.wide and i in table.wide:
            continue
        rc, ok, ms, mw = harness.wave_map_on_state(pk, table, i, reqs[p], tables=2, nic_bits=int(bits[p]))
        assert rc == 0 and ok == 3, (rc, ok, specs[p])
        G = int(reqs[p]["n_groups"])
        for f, k in (("gpu", G), ("cpu", G + 1), ("nic_numa", G), ("nic_idx", G)):
            assert mw[f][:k].tolist() == maps[p][f][:k].tolist(), (f, specs[p], mw, maps[p])
        mapped += 1
    assert mapped >= 8


# ---- a big request's mapping with lane = tuple (big_kernel.h wide_map_wave) against wide_map --------------------------------------------
@pytest.mark.parametrize("seed", range(6))
def test_big_mapping_lane_per_tuple_equals_the_one_thread_form(seed):
    """k_big_map's wavefront form - the three stages answered with lane = tuple into bit rows, every tuple's hash computed once, the CPython
    set model then run by the first lane over those tables - against wide_map as the host twin (and the device, for nodes of more than two NUMA
    nodes) runs it: return code and mapping on EVERY node of a random cluster that takes the pod, and on some that do not, for pods of 5..8
    groups in both map types.  The device text, cut out of big_kernel.h unmodified, under the 64-thread emulation."""
    from tests.test_big_core import big_spec, host_matcher
    nl = util.random_cluster(47000 + seed, 40, occupancy=0.15 if seed % 2 else 0.0)
    rng = np.random.default_rng(900 + seed)
    tops = [refmodel.make_topology(big_spec(rng, 5, 8)) for _ in range(10)]
    m = host_matcher()
    m.FindNodes(nl, tops[:1])                                           # packs the cluster, sets the dictionary
    big = np.array([m.packer.digest_big(t) for t in tops], dtype=pack.BIG_REQ)
    fits, _, exhausted = harness.big_eval(m.packer, m.engine.table, m.engine._wide_records(), big, util.CLOCK)
    assert not exhausted
    mapped = 0
    for p in range(len(tops)):
        takers = np.flatnonzero(fits[:len(nl), p])
        others = np.flatnonzero(fits[:len(nl), p] == 0)[:3]
        for v in list(takers[:6]) + list(others):
            rc, rcs, ms, mw = harness.wave_big_map(m.packer, m.engine.table, int(v), big[p])
            assert rc == 0, (p, int(v), rcs, ms, mw)
            assert rcs[0] == 1 or not fits[v, p], (p, int(v), rcs)      # (a node the scalar filters turn away may still map: wide_map asks the stages only)
            mapped += rcs[0] == 1
    assert mapped >= 6


@pytest.mark.parametrize("seed", range(3))
def test_big_mapping_lane_per_tuple_on_wide_records_of_two_sockets(seed):
    """The same on nodes the mirror carries as wide records (65..128 cores per socket) where they have at most two NUMA nodes - the
    wavefront form reads the record instead of the planes' view; records of three or four sockets keep the one-thread form (we_big_map: -2)."""
    from tests.test_big_core import big_spec, host_matcher
    nl = util.build_cluster(util.mixed_cluster_desc(48000 + seed, 30, wide_share=0.6, occupancy=0.05))
    rng = np.random.default_rng(950 + seed)
    tops = [refmodel.make_topology(big_spec(rng, 5, 7)) for _ in range(8)]
    m = host_matcher()
    m.FindNodes(nl, tops[:1])
    big = np.array([m.packer.digest_big(t) for t in tops], dtype=pack.BIG_REQ)
    wide = m.engine._wide_records()
    two = [k for k in range(len(wide)) if int(wide[k]["numa_nodes"]) <= 2]
    assert two and len(two) < len(wide)
    agreed = mapped = 0
    for p in range(len(tops)):
        for k in two[:8]:
            rc, rcs, ms, mw = harness.wave_big_map(m.packer, m.engine.table, 0, big[p], wide=wide[k:k + 1])
            assert rc == 0, (p, k, rcs, ms, mw)
            agreed += 1
            mapped += rcs[0] == 1
    k4 = next(k for k in range(len(wide)) if int(wide[k]["numa_nodes"]) > 2)
    assert harness.wave_big_map(m.packer, m.engine.table, 0, big[0], wide=wide[k4:k4 + 1])[0] == -2
    assert agreed and mapped >= 1
