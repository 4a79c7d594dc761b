#!/usr/bin/env python3
"""Why is bench.py's headline region slower than its own repeats?  The same 20 steps timed in alternating forms on one engine:
A = 5 warm-up steps + sync + reset_stats + sync in front (bench.py's timed_region), B = sync only in front (its repeats),
C = A without reset_stats, D = B with get_stats in front.  Median us per step over the rounds, per form."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nhd_amd import pack
from nhd_amd.engine import Engine
from workload import planes, refmodel, synth

spec = synth.make_cluster(4, n_nodes=65536)
pods, groups = synth.make_pods(4, n_pods=4096)
tops = [refmodel.make_topology(s) for s in pods]
pk = pack.Packer()
table = planes.planes_from_spec(pk, spec)
reqs = pk.digest_many(tops, groups)
pk.close_signatures()
eng = Engine(0)
eng.set_dictionary(pk)
eng.upload(table, global_base=0)
eng.stage(reqs)
now = spec.clock_now
for _ in range(3000):
    eng.enqueue(now)
eng.sync()


def region(form, K=20, W=5):
    if form in "AC":
        for _ in range(W):
            eng.enqueue(now)
        eng.sync()
        if form == "A":
            eng.reset_stats()
    if form == "D":
        eng.stats()
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(K):
        eng.enqueue(now)
    eng.sync()
    return (time.perf_counter() - t0) / K * 1e6


res = {f: [] for f in "ABCD"}
for r in range(40):
    for f in ("ABCD" if r % 2 == 0 else "DCBA"):
        res[f].append(region(f))
print(json.dumps({f: {"median": float(np.median(v)), "p25": float(np.percentile(v, 25)), "p75": float(np.percentile(v, 75))} for f, v in res.items()}))
