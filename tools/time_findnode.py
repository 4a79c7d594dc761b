#!/usr/bin/env python3
"""End-to-end latency of the drop-in call HipMatcher.FindNode(nl, top) (one pending pod, as the scheduler issues it):
BASELINE configs 1-4 cluster sizes, stateless and with the persistent mirror (attach); plus the C-ABI call alone."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from workload import refmodel, synth
from nhd_amd.matcher import HipMatcher

out = []
for cfg, n in ((1, 32), (2, 4096), (3, 16384), (4, 65536)):
    spec = synth.make_cluster(cfg, n_nodes=n)
    nl = spec.build_nodes()
    pods, groups = synth.make_pods(cfg, n_pods=16)
    tops = [refmodel.make_topology(s) for s in pods]
    m = HipMatcher(clock=lambda: spec.clock_now)
    rec = {"config": cfg, "nodes": n}
    if n <= 16384:
        m.FindNode(nl, tops[0])
        t0 = time.perf_counter()
        for t in tops[:2]:
            m.FindNode(nl, t)
        rec["findnode_ms_stateless_repack"] = (time.perf_counter() - t0) / 2 * 1e3
    m.attach(nl)
    m.FindNode(nl, tops[0])
    t0 = time.perf_counter()
    for t in tops:
        m.FindNode(nl, t)
    rec["findnode_ms_attached"] = (time.perf_counter() - t0) / len(tops) * 1e3
    sub = {k: v for i, (k, v) in enumerate(nl.items()) if i % 5}
    t0 = time.perf_counter()
    for t in tops:
        m.FindNode(sub, t)
    rec["findnode_ms_attached_filtered_subset"] = (time.perf_counter() - t0) / len(tops) * 1e3
    t0 = time.perf_counter()
    for t, g in zip(tops, groups):
        m.FindNodes(nl, [t], pod_groups=[g])
    rec["findnode_ms_attached_in_kernel_filter"] = (time.perf_counter() - t0) / len(tops) * 1e3
    reqs = m.packer.digest_many(tops[:1])
    m.engine.find(reqs, spec.clock_now, want_bitmap=False)
    t0 = time.perf_counter()
    for _ in range(20):
        m.engine.find(reqs, spec.clock_now, want_bitmap=False)
    rec["abi_call_ms_nhdfit_find_1_pod"] = (time.perf_counter() - t0) / 20 * 1e3
    t0 = time.perf_counter()
    for t in tops:
        m.packer.digest_many([t])
    rec["python_digest_ms"] = (time.perf_counter() - t0) / len(tops) * 1e3
    out.append(rec)
print(json.dumps(out))
