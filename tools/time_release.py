#!/usr/bin/env python3
"""Row f2: what a pod deletion costs the attached matcher.  A pending list is scheduled (one ScheduleBatch + the
bookkeeping on the node objects), then every bound pod is released with the node's own AddResourcesFromTopology
(nhd/NHDScheduler.py:203) and a FindNode follows each release (so the mirror has to be current every time): once with the
release travelling as a delta record (nhdfit_apply_deltas), once the round-1/2 way (the node re-packed in Python and
uploaded).  Stand-in node objects (tests/sched_standin.py)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nhd_amd.matcher import HipMatcher
from tests import sched_standin
from tests.delta_check import Clock
from workload import refmodel, synth

out = []
for cfg, n, P in ((4, 4096, 512), (4, 16384, 1024)):
    row = {"config": cfg, "nodes": n, "pending_pods": P}
    for mode in ("delta", "repack"):
        spec = synth.make_cluster(cfg, n_nodes=n)
        pods, groups = synth.make_pods(cfg, n_pods=P + 1)
        for p in pods:
            p["misc_smt"] = True
        clock = Clock(spec.clock_now)
        nodes = sched_standin.adopt(spec.build_nodes(), clock)
        tops = [refmodel.make_topology(p) for p in pods]
        m = HipMatcher(clock=clock)
        if mode == "repack":
            m._on_topology = lambda *a, **k: False            # the mutator wrapper then marks the node dirty: re-pack + upload
        m.attach(nodes)
        binds = sched_standin.check_pending_pods_batched(nodes, m, tops[:P], groups[:P], now=clock.t)
        probe = tops[P]
        m.FindNode(nodes, probe)
        bound = [(i, b) for i, b in enumerate(binds) if b is not None]
        t_rel = t_find = 0.0
        for i, b in bound:
            t0 = time.perf_counter()
            nodes[b].AddResourcesFromTopology(tops[i])
            t1 = time.perf_counter()
            m.FindNode(nodes, probe)
            t2 = time.perf_counter()
            t_rel += t1 - t0
            t_find += t2 - t1
        row[mode] = {"released": len(bound), "release_call_ms": t_rel * 1e3 / len(bound), "next_findnode_ms": t_find * 1e3 / len(bound),
                     "deltas": m.delta_stats["applied"], "repacked": m.delta_stats["repacked"]}
    out.append(row)
print(json.dumps(out))
