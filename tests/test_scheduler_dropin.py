"""Drop-in proof (level L3): the UNMODIFIED reference `NHDScheduler.AttemptScheduling`
(nhd/NHDScheduler.py:249-353) driven twice over identical clusters and pod sequences - once with the
reference `Matcher`, once with `HipMatcher` swapped in at `self.matcher` - must bind every pod to the same
node and leave every node in the same state (cores / GPUs / NICs / hugepages), pod after pod, with the real
commit step (`Node.SetPhysicalIdsFromMapping`) mutating the nodes in between.

Needs the reference tree (build container only).  K8s is faked; the engine under HipMatcher is the host build
of the kernel arithmetic (tests/harness), so what is exercised here is the host side of the boundary:
attach(), dirty tracking through the reference's own mutators, candidate subsets, mapping hand-over."""
import sys
import types

import numpy as np
import pytest

from workload import refmodel, synth
from nhd_amd.matcher import HipMatcher
from tests import harness


class FakeK8S:
    def __init__(self, pod_groups):
        self.pod_groups = pod_groups
        self.binds = {}
        self.events = []

    def GetPodObj(self, pod, ns): return object()
    def GeneratePodEvent(self, *a, **k): self.events.append(a[3] if len(a) > 3 else None)
    def GetCfgMap(self, pod, ns): return ("cfg", pod)
    def GetCfgType(self, pod, ns): return "triad"
    def GetPodNodeGroups(self, pod, ns): return self.pod_groups[pod]
    def AddNADToPod(self, pod, ns, nad): return True
    def AnnotatePodGpuMap(self, ns, pod, s): return True
    def AnnotatePodConfig(self, ns, pod, s): return True
    def BindPodToNode(self, pod, node, ns):
        self.binds[pod] = node
        return True


class FakeParser:
    def __init__(self, top): self.top = top
    def CfgToTopology(self, parse_net): return self.top
    def TopologyToCfg(self): return "cfg"
    def TopologyToGpuMap(self): return ""


@pytest.fixture(scope="module")
def sched_mod(ref):
    """Import nhd.NHDScheduler with its K8s-side third-party imports stubbed out."""
    # libconf / magicattr (imported by nhd.TriadCfgParser) resolve to the stand-ins under oracle/_shim
    for name in ("kubernetes", "kubernetes.client", "kubernetes.config", "kubernetes.watch", "kubernetes.client.rest"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["kubernetes"].client = sys.modules["kubernetes.client"]
    sys.modules["kubernetes"].config = sys.modules["kubernetes.config"]
    sys.modules["kubernetes"].watch = sys.modules["kubernetes.watch"]
    sys.modules["kubernetes.client"].rest = sys.modules["kubernetes.client.rest"]
    sys.modules["kubernetes.client.rest"].ApiException = type("ApiException", (Exception,), {})
    import pkg_resources
    orig = pkg_resources.get_distribution
    pkg_resources.get_distribution = lambda n: types.SimpleNamespace(version="0.0-test") if n == "nhd" else orig(n)
    import logging
    for lg in ("nhd.NHDScheduler", "nhd.K8SMgr", "nhd.TriadCfgParser"):
        logging.getLogger(lg).addHandler(logging.NullHandler())
        logging.getLogger(lg).setLevel(logging.CRITICAL + 1)
        logging.getLogger(lg).propagate = False
    import nhd.NHDScheduler as S
    yield S
    pkg_resources.get_distribution = orig


def make_scheduler(S, ref, descs, pod_groups):
    fake = FakeK8S(pod_groups)
    S.K8SMgr.GetInstance = staticmethod(lambda: fake)
    import queue
    sched = S.NHDScheduler(queue.Queue())
    sched.nodes = {d["name"]: refmodel.build_node(d, ref) for d in descs}
    return sched, fake


def node_state(n):
    return ([c.used for c in n.cores], [g.used for g in n.gpus], [(k.pods_used, tuple(k.speed_used)) for k in n.nics],
            n.mem.free_hugepages_gb)


@pytest.mark.parametrize("cfg", [3, 4, 5])
def test_attempt_scheduling_same_binds_and_state(ref, sched_mod, cfg):
    from oracle import ref_loader
    S = sched_mod
    clock = ref_loader.VirtualClock(1.0e6).install()
    spec = synth.make_cluster(cfg, n_nodes=40)
    descs = [spec.describe(i) for i in range(spec.n)]
    pods, groups = synth.make_pods(cfg, n_pods=60)
    for p in pods:
        p["misc_smt"] = True        # keep the reference's commit step out of its own buggy unwind path
    names = [f"pod{i}" for i in range(len(pods))]
    pg = dict(zip(names, groups))

    a, ka = make_scheduler(S, ref, descs, pg)
    b, kb = make_scheduler(S, ref, descs, pg)
    b.matcher = HipMatcher(clock=lambda: clock.t, engine_factory=harness.HarnessEngine)
    b.matcher.attach(b.nodes)

    placed = 0
    for name, spec_pod in zip(names, pods):
        clock.t += 1.0
        top_a = refmodel.make_topology(spec_pod, ref)
        top_b = refmodel.make_topology(spec_pod, ref)
        a.GetCfgParser = lambda t, s, _t=top_a: FakeParser(_t)
        b.GetCfgParser = lambda t, s, _t=top_b: FakeParser(_t)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            ra = a.AttemptScheduling(name, "ns")
        rb = b.AttemptScheduling(name, "ns")
        assert ra == rb, name
        assert ka.binds.get(name) == kb.binds.get(name), name
        if ra:
            placed += 1
            node = ka.binds[name]
            assert node_state(a.nodes[node]) == node_state(b.nodes[node]), (name, node)
            # the filled-in request (physical core / GPU ids, NIC MACs) is identical too
            ca = [c.core for g in top_a.proc_groups for c in g.proc_cores + g.misc_cores] + [c.core for c in top_a.misc_cores]
            cb = [c.core for g in top_b.proc_groups for c in g.proc_cores + g.misc_cores] + [c.core for c in top_b.misc_cores]
            assert ca == cb and [p.mac for p in top_a.nic_core_pairing] == [p.mac for p in top_b.nic_core_pairing]
    assert placed >= 10
    assert a.failed_schedule_count == b.failed_schedule_count
    for k in a.nodes:
        assert node_state(a.nodes[k]) == node_state(b.nodes[k]), k


class LifecycleK8S(FakeK8S):
    """The fake K8s with the pod-side calls of the release / restart paths (nhd/NHDScheduler.py:107-204)."""
    def __init__(self, pod_groups):
        super().__init__(pod_groups)
        self.gone = set()

    def GetCfgAnnotations(self, pod, ns): return False if pod in self.gone else pod       # the "config string" is the pod's name
    def GetPodNode(self, pod, ns): return self.binds.get(pod)
    def GetScheduledPods(self, sched_name): return [(p, "ns", "uid-" + p, "Running") for p in self.binds if p not in self.gone]


@pytest.mark.parametrize("cfg", [3, 4, 5])
def test_pod_lifecycle_through_the_unmodified_scheduler(ref, sched_mod, cfg):
    """Schedule, delete, schedule again, lose track (ResetResources + LoadDeployedConfigs reclaiming every running pod) -
    all through the UNMODIFIED scheduler methods (AttemptScheduling, ReleasePodResources, ResetResources: NHDScheduler.py:
    249-353, 185-204, 146-157, 107-144), once with the reference Matcher and once with HipMatcher attached.  Binds and node
    states must agree step by step; on the HipMatcher side the release / reset / reclaim calls reach the mirror as delta
    records (row f2) - the mirror equals a fresh pack of the scheduler's nodes at the end and no node was re-uploaded."""
    import contextlib, io
    from oracle import ref_loader
    from nhd_amd import pack
    S = sched_mod
    clock = ref_loader.VirtualClock(1.0e6).install()
    spec = synth.make_cluster(cfg, n_nodes=32)
    descs = [spec.describe(i) for i in range(spec.n)]
    pods, groups = synth.make_pods(cfg, n_pods=90)
    for p in pods:
        p["misc_smt"] = True
    names = [f"pod{i}" for i in range(len(pods))]
    pg = dict(zip(names, groups))

    def make():
        fake = LifecycleK8S(pg)
        S.K8SMgr.GetInstance = staticmethod(lambda: fake)
        import queue
        sched = S.NHDScheduler(queue.Queue())
        sched.nodes = {d["name"]: refmodel.build_node(d, ref) for d in descs}
        tops = {nm: refmodel.make_topology(sp, ref) for nm, sp in zip(names, pods)}
        sched.GetCfgParser = lambda t, s, _tops=tops: FakeParser(_tops[s] if s in _tops else _tops[s[1]])   # cfgstr: name, or ("cfg", name)
        return sched, fake

    a, ka = make()
    b, kb = make()
    m = HipMatcher(clock=lambda: clock.t, engine_factory=harness.HarnessEngine)
    b.matcher = m
    uploads = []
    m.attach(b.nodes)
    orig_upload = m.engine.upload
    m.engine.upload = lambda *x, **k: (uploads.append(x[0].n), orig_upload(*x, **k))[1]

    def both(fn):
        with contextlib.redirect_stdout(io.StringIO()):
            ra = fn(a)
        rb = fn(b)
        assert ra == rb
        assert ka.binds == kb.binds
        for k in a.nodes:
            assert node_state(a.nodes[k]) == node_state(b.nodes[k]), k
        return ra

    def schedule(lo, hi):
        n = 0
        for name in names[lo:hi]:
            clock.t += 1.0
            n += bool(both(lambda s: s.AttemptScheduling(name, "ns")))
        return n

    assert schedule(0, 50) >= 10
    bound = [nm for nm in names[:50] if nm in ka.binds]
    for name in bound[::2]:                                  # pods complete: their resources go back (ReleasePodResources)
        clock.t += 1.0
        both(lambda s: s.ReleasePodResources(name, "ns"))
        ka.gone.add(name); kb.gone.add(name)
    assert schedule(50, 75) >= 3                             # the freed resources are handed out again
    clock.t += 1.0
    both(lambda s: s.ReleasePodResources("pod-nobody-knows", "ns") if not (ka.gone.add("pod-nobody-knows") or kb.gone.add("pod-nobody-knows")) else None)
    # ^ a pod the API server no longer has: the scheduler resets every node and reclaims what is still running (:190-193)
    assert schedule(75, 90) >= 1
    assert m.delta_stats["applied"] >= len(bound) // 2 + len(b.nodes)
    assert len(uploads) == m.delta_stats["repacked"]
    m.FindNode(b.nodes, refmodel.make_topology(pods[0], ref))               # flush
    got = m.engine.download()
    want = m.packer.pack_nodes(b.nodes)
    for f in ("p0", "p1", "p2", "p3", "p4", "detail"):
        assert np.array_equal(getattr(got, f), getattr(want, f)), f


def test_attempt_scheduling_with_pods_of_more_than_four_groups(ref, sched_mod):
    """The same proof with pods the table pass cannot express (5..6 processing groups, nhdfit_big_req) among ordinary ones: the
    UNMODIFIED AttemptScheduling binds every pod to the reference Matcher's node and leaves the nodes in the same state; on the
    HipMatcher side the reference's own commit of such a pod is mirrored by nhdfit_big_commit (the host twin's here) - no node of
    the planes is re-packed for it - and a released big pod's resources go back as a delta."""
    import contextlib, io
    from oracle import ref_loader
    from tests import util
    from tests.test_big_core import big_spec
    S = sched_mod
    clock = ref_loader.VirtualClock(1.0e6).install()
    descs = util.random_cluster_desc(77100, 30, occupancy=0.05)
    for d in descs:
        keep, lab = 0, {}
        for k, v in d["labels"].items():
            if "nfd-extras-nic" in k:
                keep += 1
                if keep > 4:
                    continue
            lab[k] = v
        d["labels"] = lab
        d["nic_pods_used"] = [0] * sum(1 for k in lab if "nfd-extras-nic" in k and "10000Mbs" not in k.replace("100000Mbs", ""))
        d["labels"].pop("NHD_GROUP", None)
    rng = np.random.default_rng(5)
    pods = []
    for _ in range(40):
        s = big_spec(rng, 5, 6) if rng.random() < 0.5 else util.random_pod_spec(rng)
        s["misc_smt"] = True
        if s["map_type"] == "NONE":
            s["map_type"] = "NUMA"
        pods.append(s)
    names = [f"pod{i}" for i in range(len(pods))]
    pg = {nm: ["default"] for nm in names}

    def make():
        fake = LifecycleK8S(pg)
        S.K8SMgr.GetInstance = staticmethod(lambda: fake)
        import queue
        sched = S.NHDScheduler(queue.Queue())
        sched.nodes = {d["name"]: refmodel.build_node(d, ref) for d in descs}
        tops = {nm: refmodel.make_topology(sp, ref) for nm, sp in zip(names, pods)}
        sched.GetCfgParser = lambda t, s, _tops=tops: FakeParser(_tops[s] if s in _tops else _tops[s[1]])
        return sched, fake, tops

    a, ka, _ = make()
    b, kb, tops_b = make()
    m = HipMatcher(clock=lambda: clock.t, engine_factory=harness.HarnessEngine)
    b.matcher = m
    m.attach(b.nodes)
    uploads = []
    orig_upload = m.engine.upload
    m.engine.upload = lambda *x, **k: (uploads.append(x[0].n), orig_upload(*x, **k))[1]
    placed_big = 0
    for name, sp in zip(names, pods):
        clock.t += 1.0
        with contextlib.redirect_stdout(io.StringIO()):
            ra = a.AttemptScheduling(name, "ns")
        rb = b.AttemptScheduling(name, "ns")
        assert ra == rb and ka.binds.get(name) == kb.binds.get(name), name
        for k in a.nodes:
            assert node_state(a.nodes[k]) == node_state(b.nodes[k]), (name, k)
        placed_big += bool(ra) and len(sp["groups"]) > 4
    assert placed_big >= 3
    for name in [nm for nm, sp in zip(names, pods) if nm in ka.binds and len(sp["groups"]) > 4][:2]:    # a big pod completes
        clock.t += 1.0
        with contextlib.redirect_stdout(io.StringIO()):
            a.ReleasePodResources(name, "ns")
        b.ReleasePodResources(name, "ns")
        ka.gone.add(name); kb.gone.add(name)
    m.FindNode(b.nodes, tops_b[names[0]])                                     # flush
    assert uploads == [] and m.delta_stats["repacked"] == 0                   # commits and releases travelled as records, nothing was re-packed
    got, want = m.engine.download(), m.packer.pack_nodes(b.nodes)
    for f in ("p0", "p1", "p2", "p3", "p4", "detail"):
        assert np.array_equal(getattr(got, f), getattr(want, f)), f
