#!/usr/bin/env python3
"""nhdfit_find for a whole batch (host request records in, winners + mappings out) on BASELINE shapes, through ctypes: the call's wall time
(minimum and median of 9) and, with NHDFIT_FIND_PROF=1 under the tuning build, the host-side phases of each call on stderr."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from workload import planes, refmodel, synth
from nhd_amd import pack
from nhd_amd.engine import Engine

shapes = [tuple(int(v) for v in x.split(":")) for x in (sys.argv[1] if len(sys.argv) > 1 else "4:65536:4096,2:0:0,3:0:0").split(",")]
out = []
for cfg, n, P in shapes:
    spec = synth.make_cluster(cfg, n_nodes=n or None)
    pods, groups = synth.make_pods(cfg, n_pods=P or None)
    pk = pack.Packer()
    table = planes.planes_from_spec(pk, spec)
    reqs = pk.digest_many([refmodel.make_topology(s) for s in pods], groups)
    pk.close_signatures()
    eng = Engine(0)
    eng.set_dictionary(pk)
    eng.upload(table)
    now = spec.clock_now
    for _ in range(3):
        eng.find(reqs, now, want_bitmap=False, want_map=True)
    ts = []
    for _ in range(9):
        t0 = time.perf_counter()
        eng.find(reqs, now, want_bitmap=False, want_map=True)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    out.append({"config": cfg, "nodes": int(table.n), "pods": int(len(reqs)), "ms_min": ts[0] * 1e3, "ms_median": ts[len(ts) // 2] * 1e3,
                "single_launch_calls": int(eng.stats().small_finds)})
    sys.stderr.write(f"---- config {cfg} done\n")
print(json.dumps(out))
