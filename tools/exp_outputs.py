#!/usr/bin/env python3
"""Experiment: pipelined step time of the bench workload with the bitmap / mapping outputs switched off
(which roles of the step kernel cost what).  Run through gpurun."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nhd_amd import pack
from workload import planes, refmodel, synth
from nhd_amd.engine import Engine

n, P, steps = 65536, 4096, 400
spec = synth.make_cluster(4, n_nodes=n)
pods, groups = synth.make_pods(4, n_pods=P)
tops = [refmodel.make_topology(s) for s in pods]
pk = pack.Packer()
table = planes.planes_from_spec(pk, spec)
reqs = pk.digest_many(tops, groups)
eng = Engine(0)
eng.set_dictionary(pk)
eng.upload(table)
out = {}
for bitmap, mapping in ((1, 1), (1, 0), (0, 1), (0, 0)):
    eng.set_outputs(bool(bitmap), bool(mapping))
    eng.stage(reqs)
    for _ in range(20):
        eng.enqueue(spec.clock_now)
    eng.sync(); eng.reset_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.enqueue(spec.clock_now)
    eng.sync()
    dt = time.perf_counter() - t0
    st = eng.stats()
    out[f"bitmap={bitmap},map={mapping}"] = {"us_per_step": dt / steps * 1e6, "kernel_us": st.fit_ms_total / max(1, st.launches) * 1e3}
print(json.dumps(out, indent=1))
