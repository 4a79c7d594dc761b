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
