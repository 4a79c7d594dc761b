#!/usr/bin/env python3
"""Time mode B (nhdfit_schedule_batch: decide + commit on the device) on the GPU: BASELINE config-4 cluster,
sequential commit inside the batch.  Prints one JSON line."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nhd_amd import pack
from workload import planes, refmodel, synth
from nhd_amd.engine import Engine

n, P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 4
spec = synth.make_cluster(cfg, n_nodes=n)
pods, groups = synth.make_pods(cfg, n_pods=P)
tops = [refmodel.make_topology(s) for s in pods]
pk = pack.Packer(); table = planes.planes_from_spec(pk, spec); reqs = pk.digest_many(tops, groups)
added = pk.close_signatures()
eng = Engine(0); eng.set_dictionary(pk); eng.upload(table)
eng.schedule_batch(reqs, spec.clock_now, pk, apply=False)
t = []
for _ in range(5):
    t0 = time.perf_counter(); node, maps, places, status = eng.schedule_batch(reqs, spec.clock_now, pk, apply=False); t.append(time.perf_counter() - t0)
t0 = time.perf_counter(); eng.find(reqs, spec.clock_now, want_bitmap=False); ta = time.perf_counter() - t0
print(json.dumps({"config": cfg, "nodes": n, "pods": P, "mode_b_ms": min(t) * 1e3, "mode_a_call_ms": ta * 1e3, "placed": int((node >= 0).sum()),
                  "distinct_nodes": len(set(node[node >= 0].tolist())), "decisions_per_s": P / min(t), "us_per_pod": min(t) / P * 1e6,
                  "commit_would_raise": int((status == 1).sum()), "signatures": len(pk.sigs), "closure_added": added}))
