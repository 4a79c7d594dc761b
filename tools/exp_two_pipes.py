#!/usr/bin/env python3
"""Experiment (round 3): do two independent step pipelines on two HIP streams hide the per-launch fixed cost of k_step?
Two contexts on one device hold the same mirror and the same staged requests; steps alternate between them.  Prints the
step rate of one context alone and of the pair.  (If the pair wins, the library grows a second pipe natively.)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nhd_amd import pack
from nhd_amd.engine import Engine
from workload import planes, refmodel, synth

cfg, n, P = 4, 65536, 4096
spec = synth.make_cluster(cfg, n_nodes=n)
pods, groups = synth.make_pods(cfg, n_pods=P)
tops = [refmodel.make_topology(s) for s in pods]
pk = pack.Packer(); table = planes.planes_from_spec(pk, spec); reqs = pk.digest_many(tops, groups); pk.close_signatures()
engs = []
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    e = Engine(0); e.set_dictionary(pk); e.upload(table); e.stage(reqs); engs.append(e)
now = spec.clock_now
def run(active, steps):
    for k in range(50):
        active[k % len(active)].enqueue(now)
    for e in active: e.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        active[k % len(active)].enqueue(now)
    for e in active: e.sync()
    return (time.perf_counter() - t0) / steps * 1e6
out = {}
for rep in range(3):
    out.setdefault("one", []).append(run(engs[:1], 400))
    out.setdefault("pair", []).append(run(engs, 400))
s0, _, _ = engs[0].fetch(want_bitmap=False, want_map=True)
s1, _, _ = engs[-1].fetch(want_bitmap=False, want_map=True)
out["same_scores"] = bool((s0 == s1).all())
print(json.dumps(out))
