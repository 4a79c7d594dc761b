#!/usr/bin/env python3
"""Latency of nhdfit_find for ONE pod (the single-launch form) on BASELINE cluster shapes, through ctypes.
NHDFIT_LIBRARY=.../libnhdfit_tuning.so NHDFIT_ROLE_TIMES=0 prints the phases of a launch on the device clock;
NHDFIT_FIND_BLOCKS=<n> overrides the number of fit blocks."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from workload import planes, refmodel, synth
from nhd_amd import pack
from nhd_amd.engine import Engine

shapes = [(int(a), int(b)) for a, b in (x.split(":") for x in (sys.argv[1] if len(sys.argv) > 1 else "2:4096,3:16384,4:65536,5:32768").split(","))]
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
out = []
for cfg, n in shapes:
    spec = synth.make_cluster(cfg, n_nodes=n)
    pods, groups = synth.make_pods(cfg, n_pods=64)
    tops = [refmodel.make_topology(s) for s in pods]
    pk = pack.Packer()
    table = planes.planes_from_spec(pk, spec)
    reqs = pk.digest_many(tops, groups)
    reqs = reqs[reqs["n_groups"] <= 3]
    eng = Engine(0)
    eng.set_dictionary(pk)
    eng.upload(table)
    rec = {"config": cfg, "nodes": n}
    for P in (1, 64):
        for k in range(4):
            eng.find(reqs[k:k + P], spec.clock_now, want_bitmap=False)
        ts = []
        for k in range(calls):
            one = reqs[k % (len(reqs) - P + 1):][:P]
            t0 = time.perf_counter()
            eng.find(one, spec.clock_now, want_bitmap=False)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        rec[f"P{P}_ms_median"] = ts[len(ts) // 2] * 1e3
        rec[f"P{P}_ms_min"] = ts[0] * 1e3
    rec["single_launch_calls"] = int(eng.stats().small_finds)
    out.append(rec)
    eng.close()
print(json.dumps(out))
