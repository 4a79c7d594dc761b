#!/usr/bin/env python3
"""pods/s of the scheduler loop INCLUDING the bookkeeping on the node objects (row f4): the pending list of
CheckPendingPods (nhd/NHDScheduler.py:425-437) served pod by pod (FindNode + commit mirrored on the device per pod) and as
one device pass (ScheduleBatch) followed by the same bookkeeping.  Stand-in node objects (tests/sched_standin.py): the
bookkeeping is the oracle's Python commit step, i.e. about what the reference's own SetPhysicalIdsFromMapping costs."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import sched_check
from workload import synth

out = []
for cfg, n, P in ((4, 4096, 512), (4, 16384, 1024)):
    case = {"config": cfg, "n_nodes": n, "n_pods": P, "clock0": synth.make_cluster(cfg, n_nodes=n).clock_now, "dt": 0.0}
    row = {"config": cfg, "nodes": n, "pending_pods": P}
    for batched in (False, True):
        nodes, m, binds, extra = sched_check.replay(case, batched=batched)
        row["batched" if batched else "pod_by_pod"] = {"bound": sum(b is not None for b in binds), "loop_seconds": m.loop_seconds,
                                                       "pods_per_s": P / m.loop_seconds, "nodes_re_uploaded": extra}
    # pod by pod like the first leg, but the scheduler passes the pod's groups instead of building the filtered dict
    # (`FindNodes(self.nodes, [top], pod_groups=[groups])[0]` in place of NHDScheduler.py:274-277)
    import time
    from nhd_amd.matcher import HipMatcher
    from tests import sched_standin
    from workload import refmodel
    spec = synth.make_cluster(cfg, n_nodes=n)
    pods, groups = synth.make_pods(cfg, n_pods=P)
    for p in pods:
        p["misc_smt"] = True
    clock = sched_check.Clock(case["clock0"])
    nodes = sched_standin.adopt(spec.build_nodes(), clock)
    tops = [refmodel.make_topology(p) for p in pods]
    m = HipMatcher(clock=clock)
    m.attach(nodes)
    t0 = time.perf_counter()
    bound = 0
    for top, grp in zip(tops, groups):
        match = m.FindNodes(nodes, [top], pod_groups=[grp])[0]
        bound += sched_standin.attempt_scheduling(nodes, m, top, grp, match=match) is not None
    dt = time.perf_counter() - t0
    row["pod_by_pod_in_kernel_filter"] = {"bound": bound, "loop_seconds": dt, "pods_per_s": P / dt}
    out.append(row)
print(json.dumps(out))
