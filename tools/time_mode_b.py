#!/usr/bin/env python3
"""Time mode B (nhdfit_find_sequential) on the GPU: BASELINE config-4 cluster, sequential commit inside the batch."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nhd_amd import pack, refmodel, synth
from nhd_amd.engine import Engine

n, P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, int(sys.argv[2]) if len(sys.argv) > 2 else 4096
spec = synth.make_cluster(4, n_nodes=n)
pods, groups = synth.make_pods(4, n_pods=P)
for p in pods:
    p["misc_smt"] = True
tops = [refmodel.make_topology(s) for s in pods]
pk = pack.Packer(); table = pk.planes_from_spec(spec); reqs = pk.digest_many(tops, groups)
eng = Engine(0); eng.set_dictionary(pk); eng.upload(table)
eng.find_sequential(reqs, spec.clock_now)
t = []
for _ in range(5):
    t0 = time.perf_counter(); node, maps, status = eng.find_sequential(reqs, spec.clock_now); t.append(time.perf_counter() - t0)
t0 = time.perf_counter(); eng.find(reqs, spec.clock_now, want_bitmap=False); ta = time.perf_counter() - t0
print(json.dumps({"nodes": n, "pods": P, "mode_b_ms": min(t) * 1e3, "mode_a_call_ms": ta * 1e3, "placed": int((node >= 0).sum()),
                  "distinct_nodes": len(set(node[node >= 0].tolist())), "decisions_per_s": P / min(t), "commit_failures": int(status.sum())}))
