"""HipMatcher's HOST logic (packing, persistent mirror + dirty tracking, candidate masks, decoding) on CPU,
with the host build of the kernel arithmetic injected as the engine (tests/harness.HarnessEngine)."""
import numpy as np
import pytest

from workload import refmodel, synth
from nhd_amd.matcher import HipMatcher
from oracle import nhd_oracle as O
from tests import harness, util


def norm(res):
    return (None,) if res[0] is None else (res[0], {"gpu": tuple(res[1]["gpu"]), "cpu": tuple(res[1]["cpu"]),
                                                    "nic": [tuple(x) for x in res[1]["nic"]]})


def test_stateless_findnode_matches_oracle():
    nl = util.random_cluster(99, 60)
    m = HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine)
    rng = np.random.default_rng(1)
    for _ in range(40):
        top = refmodel.make_topology(util.random_pod_spec(rng))
        assert m.FindNode(nl, top) == norm(O.find_node(nl, top, util.CLOCK))
    assert m.FindNode({}, top) == (None,)


def test_attach_mode_dirty_tracking_and_candidate_subsets():
    spec = synth.make_cluster(4, n_nodes=200)
    nl = spec.build_nodes()
    pods, groups = synth.make_pods(4, n_pods=30)
    tops = [refmodel.make_topology(s) for s in pods]
    m = HipMatcher(clock=lambda: spec.clock_now, engine_factory=harness.HarnessEngine)
    m.attach(nl)
    first = m.FindNodes(nl, tops)
    assert [norm(r) for r in first] == [norm(O.find_node(nl, t, spec.clock_now)) for t in tops]
    winners = list(dict.fromkeys(r[0] for r in first if r[0] is not None))
    others = [k for k in nl if k not in winners]
    a, b, c = (winners + others)[:3]
    nl[a].maintenance = True                          # watched attribute -> tracked automatically
    nl[b].busy_time = spec.clock_now
    for g in nl[c].gpus:
        g.used = True
    for k in nl[c].cores:
        k.used = True
    m.mark_dirty(c)                                   # inner object write -> explicit mark
    assert len(m._dirty) == 3
    second = m.FindNodes(nl, tops)
    assert not m._dirty
    assert [norm(r) for r in second] == [norm(O.find_node(nl, t, spec.clock_now)) for t in tops]
    sub = {k: v for i, (k, v) in enumerate(nl.items()) if i % 4 != 1}
    for t in tops[:12]:
        assert m.FindNode(sub, t) == norm(O.find_node(sub, t, spec.clock_now))
    with pytest.raises(ValueError):
        m.FindNode(dict(reversed(list(nl.items()))), tops[0])       # order differs from the attached dict


def test_pod_groups_filter_equals_scheduler_side_filter():
    spec = synth.make_cluster(5, n_nodes=300)
    nl = spec.build_nodes()
    pods, groups = synth.make_pods(5, n_pods=25)
    tops = [refmodel.make_topology(s) for s in pods]
    m = HipMatcher(clock=lambda: spec.clock_now, engine_factory=harness.HarnessEngine)
    got = m.FindNodes(nl, tops, pod_groups=groups)
    for top, grp, res in zip(tops, groups, got):
        sub = O.initial_node_filter(nl, grp)
        assert norm(res) == norm(O.find_node(sub, top, spec.clock_now))


def test_attached_dict_grows_and_foreign_dicts_do_not_corrupt_the_mirror():
    """a node added to the attached dict must be hooked and mirrored; a call with a dict the mirror does
    not know (names outside the attached dict) runs stateless and must not leave a subset behind as 'the mirror'."""
    spec = synth.make_cluster(3, n_nodes=120)
    every = spec.build_nodes()
    names = list(every)
    nl = {k: every[k] for k in names[:100]}
    pods, _ = synth.make_pods(3, n_pods=20)
    tops = [refmodel.make_topology(s) for s in pods]
    m = HipMatcher(clock=lambda: spec.clock_now, engine_factory=harness.HarnessEngine)
    m.attach(nl)
    m.FindNodes(nl, tops)
    for k in names[100:110]:                           # the scheduler learns about new nodes (BuildInitialNodeList again)
        nl[k] = every[k]
    assert [norm(r) for r in m.FindNodes(nl, tops)] == [norm(O.find_node(nl, t, spec.clock_now)) for t in tops]
    nl[names[105]].maintenance = True                  # a node that joined later is tracked like the others
    assert names[105] in m._dirty
    assert [norm(r) for r in m.FindNodes(nl, tops)] == [norm(O.find_node(nl, t, spec.clock_now)) for t in tops]
    foreign = {k: every[k] for k in names[110:]}       # not part of the attached dict at all
    for t in tops[:5]:
        assert norm(m.FindNode(foreign, t)) == norm(O.find_node(foreign, t, spec.clock_now))
    assert [norm(r) for r in m.FindNodes(nl, tops)] == [norm(O.find_node(nl, t, spec.clock_now)) for t in tops]
    nl[names[3]].maintenance = True
    assert [norm(r) for r in m.FindNodes(nl, tops)] == [norm(O.find_node(nl, t, spec.clock_now)) for t in tops]


def test_reattach_after_node_churn_compacts_the_dictionary():
    """Nodes leave and join the attached dict: the matcher re-attaches with a FRESH packer - signatures / group sets only
    the departed nodes had are gone, results stay the oracle's, and pods given as config-less requests with groups (the
    in-kernel InitialNodeFilter) use the new group ids."""
    spec = synth.make_cluster(5, n_nodes=160)
    nl = spec.build_nodes()
    pods, groups = synth.make_pods(5, n_pods=40)
    tops = [refmodel.make_topology(s) for s in pods]
    m = HipMatcher(clock=lambda: spec.clock_now, engine_factory=harness.HarnessEngine)
    m.attach(nl)
    want = [norm(O.find_node(O.initial_node_filter(nl, g), t, spec.clock_now)) for t, g in zip(tops, groups)]
    assert [norm(r) for r in m.FindNodes(nl, tops, pod_groups=groups)] == want
    sigs_before, sets_before, packer_before = len(m.packer.sigs), len(m.packer.group_sets), m.packer
    for name in list(nl)[:120]:                      # three quarters of the cluster go away ...
        del nl[name]
    extra = synth.make_cluster(5, n_nodes=200).build_nodes()
    for name in list(extra)[160:170]:                # ... ten new nodes join
        nl[name] = extra[name]
    want = [norm(O.find_node(O.initial_node_filter(nl, g), t, spec.clock_now)) for t, g in zip(tops, groups)]
    assert [norm(r) for r in m.FindNodes(nl, tops, pod_groups=groups)] == want
    assert m.packer is not packer_before
    assert len(m.packer.sigs) <= sigs_before and len(m.packer.group_sets) < sets_before
    assert len(m._names) == 50


def _odd_cluster():
    """A cluster with nodes the device layout cannot hold next to ordinary ones: four sockets, 96 physical cores per socket."""
    from workload.refmodel import NFD
    nl = util.random_cluster(4321, 24)
    base = {"DATA_PLANE_VLAN": "1", "DATA_DEFAULT_GW": "10.0.0.1/32",
            NFD + "nfd-extras-nic.eth0.mlx.0000000000aa.100000Mbs.0.10.0.0": "true",
            NFD + "nfd-extras-nic.eth1.mlx.0000000000bb.100000Mbs.1.20.1.0": "true"}
    quad = dict(base, **{NFD + "nfd-extras-cpu.numSockets": "4", NFD + "nfd-extras-cpu.num_cores": "64"})
    wide = dict(base, **{NFD + "nfd-extras-cpu.numSockets": "2", NFD + "nfd-extras-cpu.num_cores": "192",
                         NFD + "cpu-hardware_multithreading": "true"})
    penta = {NFD + "nfd-extras-cpu.numSockets": "5", NFD + "nfd-extras-cpu.num_cores": "40",
             NFD + "nfd-extras-nic.eth0.mlx.0000000000c0.100000Mbs.0.10.0.0": "true",
             NFD + "nfd-extras-nic.eth1.mlx.0000000000c1.100000Mbs.4.20.1.0": "true",
             "DATA_PLANE_VLAN": "7", "DATA_DEFAULT_GW": "10.1.0.1/32"}
    out = {}
    for k, (name, node) in enumerate(nl.items()):
        if k == 1:
            out["penta-socket"] = refmodel.node_from_labels("penta-socket", penta, (64, 64))
        if k == 3:
            out["quad-socket"] = refmodel.node_from_labels("quad-socket", quad, (64, 64))
        if k == 9:
            out["wide-socket"] = refmodel.node_from_labels("wide-socket", wide, (64, 64))
        out[name] = node
    return out


def test_nodes_beyond_the_layouts_never_match_wide_nodes_do_and_nothing_raises(caplog):
    """"failure is (None,) - never an exception".  A node beyond the fast layout (four sockets, 96 physical
    cores per socket) is served by the general path: FindNode's answers equal the oracle's on the cluster WITH them.  A node
    no layout holds (five sockets) is left out - every other node is answered for exactly - and is named in `unmirrored` and in
    the log."""
    nl = _odd_cluster()
    supported = {k: v for k, v in nl.items() if k != "penta-socket"}
    rng = np.random.default_rng(5)
    on_wide = 0
    for attach in (False, True):
        m = HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine)
        if attach:
            m.attach(nl)
        with caplog.at_level("WARNING"):
            for _ in range(25):
                top = refmodel.make_topology(util.random_pod_spec(rng))
                got = m.FindNode(nl, top)
                assert got == norm(O.find_node(supported, top, util.CLOCK))
                on_wide += got[0] in ("quad-socket", "wide-socket")
        assert set(m.unmirrored) == {"penta-socket"} and set(m.wide_nodes) == {"quad-socket", "wide-socket"}
        assert "5 NUMA nodes" in m.unmirrored["penta-socket"]
    assert on_wide > 0
    assert "penta-socket" in caplog.text and "never be selected" in caplog.text
    with pytest.raises(Exception):
        HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine, strict=True).FindNode(nl, top)


def test_requests_beyond_the_record_and_device_errors_answer_none(caplog):
    nl = util.random_cluster(99, 20)
    m = HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine)
    ok = refmodel.make_topology(dict(map_type="NUMA", hugepages_gb=0, misc=0, misc_smt=True,
                                     groups=[dict(proc=2, helpers=0, rx=0, tx=0, gpus=[], proc_smt=False, helper_smt=False)]))
    five = refmodel.make_topology(dict(map_type="NUMA", hugepages_gb=0, misc=0, misc_smt=True,
                                       groups=[dict(proc=2, helpers=0, rx=0, tx=0, gpus=[], proc_smt=False, helper_smt=False)] * 5))
    huge = refmodel.make_topology(dict(map_type="NUMA", hugepages_gb=5000, misc=0, misc_smt=True,
                                       groups=[dict(proc=2, helpers=0, rx=0, tx=0, gpus=[], proc_smt=False, helper_smt=False)]))
    nine = refmodel.make_topology(dict(map_type="NUMA", hugepages_gb=0, misc=0, misc_smt=True,
                                       groups=[dict(proc=2, helpers=0, rx=0, tx=0, gpus=[], proc_smt=False, helper_smt=False)] * 9))
    with caplog.at_level("ERROR"):
        got = m.FindNodes(nl, [ok, five, huge, ok, nine])
    # five groups, or more hugepages than the pod tile's table holds: answered by the general path (nhdfit_big_req), as the reference would
    assert got[0] == norm(O.find_node(nl, ok, util.CLOCK)) == got[3] and got[1] == norm(O.find_node(nl, five, util.CLOCK)) and got[1][0] is not None
    assert got[2] == norm(O.find_node(nl, huge, util.CLOCK)) == (None,)
    assert got[4] == (None,) and "cannot be expressed" in caplog.text              # nine groups: beyond every record
    roomy = refmodel.make_topology(dict(map_type="NUMA", hugepages_gb=2000, misc=0, misc_smt=True,
                                        groups=[dict(proc=2, helpers=0, rx=0, tx=0, gpus=[], proc_smt=False, helper_smt=False)]))
    big_mem = dict(nl)
    last = list(big_mem)[-1]
    big_mem[last].mem.free_hugepages_gb = 4096                                      # a node with terabytes of 1G pages takes the 2 000 GiB pod
    assert norm(m.FindNode(big_mem, roomy)) == norm(O.find_node(big_mem, roomy, util.CLOCK)) and m.FindNode(big_mem, roomy)[0] is not None

    from nhd_amd._lib import NhdFitError

    class Broken(harness.HarnessEngine):
        def find(self, *a, **kw):
            raise NhdFitError(-3, "hipErrorLaunchFailure (injected)")
    m2 = HipMatcher(clock=lambda: util.CLOCK, engine_factory=Broken)
    with caplog.at_level("ERROR"):
        assert m2.FindNode(nl, ok) == (None,)
    assert "device path failed" in caplog.text
    with pytest.raises(NhdFitError):
        HipMatcher(clock=lambda: util.CLOCK, engine_factory=Broken, strict=True).FindNode(nl, ok)


def test_batch_placements_never_applied_do_not_stay_in_the_mirror():
    """ScheduleBatch(apply=True) commits on the device; if the caller never applies those placements to its node
    objects, the next call must see the objects' state again (the nodes are re-packed), not the orphaned commits."""
    spec = synth.make_cluster(3, n_nodes=48)
    nl = spec.build_nodes()
    pods, groups = synth.make_pods(3, n_pods=24)
    tops = [refmodel.make_topology(s) for s in pods]
    m = HipMatcher(clock=lambda: spec.clock_now, engine_factory=harness.HarnessEngine)
    m.attach(nl)
    before = [norm(O.find_node(nl, t, spec.clock_now)) for t in tops]
    got = m.ScheduleBatch(nl, tops, apply=True)
    assert sum(r[0] is not None for r in got) >= 5 and m._batch_ids
    assert [norm(r) for r in m.FindNodes(nl, tops)] == before          # objects untouched -> same answers as before the batch
    assert not m._batch_ids


def test_reference_fixture_with_nodes_beyond_the_layout():
    """tests/golden/beyond: the unmodified reference's answers on a cluster with a four-socket node and a 96-core-per-socket node."""
    from tests import beyond_check

    def unpack(bm, n):
        chunks, P = bm.shape
        bits = np.unpackbits(bm.view(np.uint8).reshape(chunks, P, 8), axis=2, bitorder="little")
        return bits.transpose(1, 0, 2).reshape(P, chunks * 64)[:, :n]

    beyond_check.check(lambda clock: HipMatcher(clock=lambda: clock, engine_factory=harness.HarnessEngine), unpack)



def test_enable_sharing_flipped_in_the_nodes_module_changes_the_arithmetic(monkeypatch):
    """nhd/Node.py:20 ENABLE_SHARING = True makes GetFreeNumaNicResources price a NIC per direction at speed * 0.9 - speed_used[x]
    (Node.py:290).  Rounds 3-4 refused such clusters; round 5 answers them through the general path (every node a wide record with
    its NICs' speed_used, pack.WIDE_SHARE).  The constants are read from the module the node objects come from, so flipping that
    module's switch changes the answers - to the oracle's with ITS switch flipped (pinned to the reference with the constant
    flipped by tests/golden/sharing and tests/test_sharing.py) - and flipping it back restores the shipped ones;
    NIC_BW_AVAIL_PERCENT of that module is the one the capacities are computed with."""
    from nhd_amd import pack
    nl = util.random_cluster(4242, 24)
    for k, node in enumerate(nl.values()):                         # traffic on the NICs, and a pod on some of them: the two arithmetics disagree
        for j, nic in enumerate(node.nics):
            nic.speed_used = [float((7 * k + 3 * j) % 60), float((5 * k + j) % 50)]
            nic.pods_used = (k + j) % 2
    top = refmodel.make_topology(dict(map_type="NUMA", hugepages_gb=0, misc=0, misc_smt=True,
                                      groups=[dict(proc=2, helpers=0, rx=40.0, tx=45.0, gpus=[], proc_smt=False, helper_smt=False)]))
    m = HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine)
    shipped = norm(O.find_node(nl, top, util.CLOCK))
    assert m.FindNode(nl, top) == shipped
    assert pack.node_module_constants(next(iter(nl.values()))) == {"NIC_BW_AVAIL_PERCENT": 0.9, "SCHEDULABLE_NIC_SPEED_THRESH_MBPS": 11000,
                                                                  "ENABLE_SHARING": False}
    monkeypatch.setattr(refmodel, "ENABLE_SHARING", True)
    monkeypatch.setattr(O, "ENABLE_SHARING", True)
    shared = norm(O.find_node(nl, top, util.CLOCK))
    assert shared != shipped                                       # (the case tells the two apart)
    assert m.FindNode(nl, top) == shared
    assert len(m.wide_nodes) == len(nl) and m.unmirrored == {}
    m.attach(nl)                                                   # tracked subclasses still resolve to the nodes' own module
    assert m.FindNodes(nl, [top, top]) == [shared, shared]
    m.detach()
    # a processing group with two RX cores: the reference's commit adds their speeds to speed_used one after the other (nhd/Node.py:754),
    # the request record carries their sum - answered while every speed in sight is a multiple of 2^-20 Gb/s (all sums exact: round 6,
    # tests/test_sharing.py), turned away, loudly, when one is not
    def two_rx_cores(second):
        t = refmodel.make_topology(dict(map_type="NUMA", hugepages_gb=0, misc=0, misc_smt=True,
                                        groups=[dict(proc=3, helpers=0, rx=10.0, tx=5.0, gpus=[], proc_smt=False, helper_smt=False)]))
        extra = t.proc_groups[0].proc_cores[2]
        extra.nic_dir, extra.nic_speed = t.proc_groups[0].proc_cores[0].nic_dir, second
        return t
    assert m.FindNode(nl, two_rx_cores(10.0)) == norm(O.find_node(nl, two_rx_cores(10.0), util.CLOCK)) != (None,)
    assert m.FindNode(nl, two_rx_cores(0.1)) == (None,)
    with pytest.raises(pack.UnsupportedNode):
        HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine, strict=True).FindNode(nl, two_rx_cores(0.1))
    monkeypatch.setattr(refmodel, "ENABLE_SHARING", False)
    monkeypatch.setattr(O, "ENABLE_SHARING", False)
    assert m.FindNode(nl, top) == shipped                          # switched back: the shipped arithmetic again
    assert m.wide_nodes == []
    # the head-room constant is read from the same module: at 50 % a 2 x 50 Gb/s request no longer fits a 100 GbE NIC (cap 50.0)
    big = refmodel.make_topology(dict(map_type="NUMA", hugepages_gb=0, misc=0, misc_smt=True,
                                      groups=[dict(proc=2, helpers=0, rx=60.0, tx=60.0, gpus=[], proc_smt=False, helper_smt=False)]))
    before = HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine).FindNode(nl, big)
    monkeypatch.setattr(refmodel, "NIC_BW_AVAIL_PERCENT", 0.5)
    after = HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine).FindNode(nl, big)
    assert before[0] is not None and after == (None,)            # (the cluster's fastest NICs are 100 GbE: 90.0 fits 60, 50.0 does not)


def test_filtered_dicts_of_several_node_groups_taking_turns():
    """Pods of different node groups alternate in the pending list, each with its own filtered dict (InitialNodeFilter builds a
    new dict object per pod): the candidate masks of the subsets seen lately are all kept - a subset is computed once, recognised
    afterwards - and the answers are the oracle's on every call."""
    descs = util.random_cluster_desc(4711, 90, occupancy=0.1)
    for i, d in enumerate(descs):
        d["labels"]["NHD_GROUP"] = ("alpha", "beta", "alpha.gamma")[i % 3]
    nl = util.build_cluster(descs)
    rng = np.random.default_rng(8)
    tops = [refmodel.make_topology(util.random_pod_spec(rng)) for _ in range(24)]
    m = HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine)
    m.attach(nl)
    groups = (["alpha"], ["beta"], ["gamma"], ["alpha", "beta"])
    for k, top in enumerate(tops):
        sub = O.initial_node_filter(nl, groups[k % 4])                     # a fresh dict object every time
        assert m.FindNode(sub, top) == norm(O.find_node(sub, top, util.CLOCK))
    assert m._subset_count == 4 and sum(len(v) for v in m._last_subset.values()) == 4
    nl[list(nl)[3]].active = False                                         # the filter's outcome changes: a fifth subset, the others stay
    sub = O.initial_node_filter(nl, ["alpha"])
    assert m.FindNode(sub, tops[0]) == norm(O.find_node(sub, tops[0], util.CLOCK))
    assert m._subset_count == 5
