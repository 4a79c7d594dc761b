#!/usr/bin/env python3
"""The driver's timed region - sync, K enqueues, sync - repeated R times in one process on bench.py's default workload (config 4,
65 536 nodes x 4 096 pods): median / mean / min / max of the per-step time over the regions, after a settle phase.  For A/B runs of
library builds and tuning knobs (NHDFIT_LIBRARY, NHDFIT_PIPES, ...): one region is ~330 us, a single one is too noisy to compare.
    python tools/time_driver_form.py [K=20] [R=60]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nhd_amd import pack
from nhd_amd.engine import Engine
from workload import planes, refmodel, synth

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
R = int(sys.argv[2]) if len(sys.argv) > 2 else 60
spec = synth.make_cluster(4, n_nodes=65536)
pods, groups = synth.make_pods(4, n_pods=4096)
tops = [refmodel.make_topology(s) for s in pods]
pk = pack.Packer()
table = planes.planes_from_spec(pk, spec)
reqs = pk.digest_many(tops, groups)
pk.close_signatures()
eng = Engine(0)
eng.set_dictionary(pk)
eng.upload(table)
eng.stage(reqs)
now = spec.clock_now
for _ in range(3000):
    eng.enqueue(now)
eng.sync()
ts = []
for _ in range(R):
    for _ in range(5):
        eng.enqueue(now)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(K):
        eng.enqueue(now)
    eng.sync()
    ts.append((time.perf_counter() - t0) * 1e6 / K)
t0 = time.perf_counter()
for _ in range(2000):
    eng.enqueue(now)
eng.sync()
steady = (time.perf_counter() - t0) * 1e6 / 2000
ts = np.array(ts)
print(json.dumps({"steps": K, "regions": R, "us_per_step_median": float(np.median(ts)), "mean": float(ts.mean()), "min": float(ts.min()), "max": float(ts.max()),
                  "p25": float(np.percentile(ts, 25)), "p75": float(np.percentile(ts, 75)), "steady_us_per_step": steady,
                  "library": os.environ.get("NHDFIT_LIBRARY", "libnhdfit.so"), "pipes": os.environ.get("NHDFIT_PIPES")}))
