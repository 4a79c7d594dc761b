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
    out.append(row)
print(json.dumps(out))
