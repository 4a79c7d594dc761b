#!/usr/bin/env python3
"""End-to-end latency of the drop-in call HipMatcher.FindNode(nl, top) (one pending pod, as the scheduler issues it):
BASELINE config 1 (32 nodes) and config 2 (4 096 nodes), stateless and with the persistent mirror (attach)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from workload import refmodel, synth
from nhd_amd.matcher import HipMatcher

out = []
for cfg, n in ((1, 32), (2, 4096), (3, 16384)):
    spec = synth.make_cluster(cfg, n_nodes=n)
    nl = spec.build_nodes()
    pods, groups = synth.make_pods(cfg, n_pods=16)
    tops = [refmodel.make_topology(s) for s in pods]
    m = HipMatcher(clock=lambda: spec.clock_now)
    m.FindNode(nl, tops[0])
    t0 = time.perf_counter()
    for t in tops[:4]:
        m.FindNode(nl, t)
    stateless = (time.perf_counter() - t0) / 4
    m.attach(nl)
    m.FindNode(nl, tops[0])
    t0 = time.perf_counter()
    for t in tops:
        m.FindNode(nl, t)
    attached = (time.perf_counter() - t0) / len(tops)
    sub = {k: v for i, (k, v) in enumerate(nl.items()) if i % 5}
    t0 = time.perf_counter()
    for t in tops:
        m.FindNode(sub, t)
    filtered = (time.perf_counter() - t0) / len(tops)
    out.append({"config": cfg, "nodes": n, "findnode_ms_stateless_repack": stateless * 1e3, "findnode_ms_attached": attached * 1e3,
                "findnode_ms_attached_filtered_subset": filtered * 1e3})
print(json.dumps(out))
