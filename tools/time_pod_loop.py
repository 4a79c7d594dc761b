#!/usr/bin/env python3
"""The scheduler's pod-at-a-time loop through the drop-in class (bench.py sched_loop, leg pod_by_pod_kernel_filter), call by call:
FindNodes([top], pod_groups) and CommitPlacement timed separately, then the C-ABI calls under them alone (nhdfit_find with one
pod, nhdfit_commit), at BASELINE config 4's node mix.   tools/time_pod_loop.py [nodes] [pods]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from workload import refmodel, synth
from nhd_amd.matcher import HipMatcher

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
spec = synth.make_cluster(4, n_nodes=n)
pods, groups = synth.make_pods(4, n_pods=P)
tops = [refmodel.make_topology(s) for s in pods]
now = spec.clock_now
out = {"nodes": n, "pods": P}
for rep in range(2):
    nodes = spec.build_nodes()
    m = HipMatcher(clock=lambda: now)
    m.attach(nodes)
    t_find = t_commit = 0.0
    placed = 0
    t_all = time.perf_counter()
    for k, top in enumerate(tops):
        t0 = time.perf_counter()
        r = m.FindNodes(nodes, [top], pod_groups=[groups[k]])[0]
        t1 = time.perf_counter()
        if r[0] is not None:
            m.CommitPlacement(r[0], top, r[1], busy_time=now)
            placed += 1
        t2 = time.perf_counter()
        t_find += t1 - t0
        t_commit += t2 - t1
    t_all = time.perf_counter() - t_all
    out["loop_us_per_pod"] = t_all / P * 1e6
    out["findnodes_us"] = t_find / P * 1e6
    out["commitplacement_us"] = t_commit / max(placed, 1) * 1e6
    out["placed"] = placed
    if rep == 0:
        m.engine.close()
# the ABI calls alone, on the mirror the loop left: one-pod find, then find + commit alternating (what the loop issues)
reqs = [m.packer.digest_many([t], [g]) for t, g in zip(tops[:200], groups[:200])]
eng = m.engine
t0 = time.perf_counter()
for rq in reqs:
    eng.find(rq, now, want_bitmap=False)
out["abi_find_1_pod_us"] = (time.perf_counter() - t0) / len(reqs) * 1e6
tf = tc = 0.0
done = 0
for rq in reqs:
    t0 = time.perf_counter()
    score, _, maps = eng.find(rq, now, want_bitmap=False)
    t1 = time.perf_counter()
    tf += t1 - t0
    if int(score[0]):
        node = int((np.uint64(0x7FFFFFFFFFFFFFFF) - (score[0] & np.uint64(0x7FFFFFFFFFFFFFFF))))
        t1 = time.perf_counter()
        eng.commit(node, rq[0], maps[0], now)
        tc += time.perf_counter() - t1
        done += 1
out["abi_find_between_commits_us"] = tf / len(reqs) * 1e6
out["abi_commit_us"] = tc / max(done, 1) * 1e6
out["abi_commits"] = done
print(json.dumps(out))
