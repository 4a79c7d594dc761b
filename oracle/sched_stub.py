"""Import the UNMODIFIED reference scheduler module (nhd/NHDScheduler.py) in this container: its K8s-side third-party
imports (`kubernetes`, and `libconf` / `magicattr` through nhd.TriadCfgParser) are absent, so they are stubbed /
resolved to oracle/_shim, and `pkg_resources.get_distribution("nhd")` (NHDScheduler.py:57) is answered.  K8s itself is
faked (FakeK8S): binds are recorded instead of sent.  TEST INFRASTRUCTURE (build container only)."""
import logging
import sys
import types

from oracle import ref_loader


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


_mod = None


def load_scheduler_module():
    global _mod
    if _mod is not None:
        return _mod
    ref_loader.load()
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
    for lg in ("nhd.NHDScheduler", "nhd.K8SMgr", "nhd.TriadCfgParser"):
        logging.getLogger(lg).addHandler(logging.NullHandler())
        logging.getLogger(lg).setLevel(logging.CRITICAL + 1)
        logging.getLogger(lg).propagate = False
    try:
        import nhd.NHDScheduler as S
    finally:
        pkg_resources.get_distribution = orig
    _mod = S
    return S


def make_scheduler(S, nodes, pod_groups):
    """A reference NHDScheduler over `nodes` (Dict[str, reference Node]) with the fake K8s behind it."""
    import queue
    import pkg_resources
    fake = FakeK8S(pod_groups)
    S.K8SMgr.GetInstance = staticmethod(lambda: fake)
    orig = pkg_resources.get_distribution
    pkg_resources.get_distribution = lambda n: types.SimpleNamespace(version="0.0-test") if n == "nhd" else orig(n)
    try:
        sched = S.NHDScheduler(queue.Queue())
    finally:
        pkg_resources.get_distribution = orig
    sched.nodes = nodes
    return sched, fake
