#!/usr/bin/env python3
"""Cost of the general path for requests (nhdfit_big_find / nhdfit_big_commit: pods with 5..8 processing groups) on BASELINE
cluster shapes, through ctypes: one big pod per call by group count, a batch of 16 in one call, and a find + commit pair.
Run under `rocprofv3 --kernel-trace --stats` for k_big_eval / k_big_map / k_big_commit's own durations (tools/r04_big_prof.sh)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from workload import planes, refmodel, synth
from nhd_amd import pack
from nhd_amd.engine import Engine, winner_index

shapes = [(int(a), int(b)) for a, b in (x.split(":") for x in (sys.argv[1] if len(sys.argv) > 1 else "4:65536,5:32768").split(","))]
out = []
for cfg, n in shapes:
    spec = synth.make_cluster(cfg, n_nodes=n)
    pods, groups = synth.make_pods(cfg, n_pods=400)
    pk = pack.Packer()
    table = planes.planes_from_spec(pk, spec)
    pk.close_signatures()
    eng = Engine(0)
    eng.set_dictionary(pk)
    eng.upload(table)
    by_g = {}
    k = 0
    while k + 4 <= len(pods) and min(len(by_g.get(g, ())) for g in (5, 6, 7, 8)) < 6:
        for take in (2, 3, 4):
            gs = [g for s in pods[k:k + take] for g in s["groups"]]
            if 5 <= len(gs) <= 8 and len(by_g.setdefault(len(gs), [])) < 6:
                by_g[len(gs)].append(pk.digest_big(refmodel.make_topology(dict(pods[k], groups=gs)), groups[k]))
        k += 1
    rec = {"config": cfg, "nodes": n}
    every = []
    for G in sorted(by_g):
        ts, placed = [], 0
        for r in by_g[G]:
            one = np.array([r], dtype=pack.BIG_REQ)
            eng.big_find(one, spec.clock_now)
            t0 = time.perf_counter()
            sc, _ = eng.big_find(one, spec.clock_now)
            ts.append(time.perf_counter() - t0)
            placed += int(sc[0] != 0)
            every.append(r)
        ts.sort()
        rec[f"G{G}"] = {"pods": len(ts), "placed": placed, "ms_median": ts[len(ts) // 2] * 1e3, "ms_min": ts[0] * 1e3, "ms_max": ts[-1] * 1e3}
    batch = np.array(every[:16], dtype=pack.BIG_REQ)
    eng.big_find(batch, spec.clock_now)
    t0 = time.perf_counter()
    sc, mp = eng.big_find(batch, spec.clock_now)
    rec["batch16_ms"] = (time.perf_counter() - t0) * 1e3
    done = 0
    t0 = time.perf_counter()
    for j in range(len(batch)):                                   # the scheduler's loop for such pods: find, commit, next
        s1, m1 = eng.big_find(batch[j:j + 1], spec.clock_now)
        if s1[0]:
            eng.big_commit(winner_index(int(s1[0])), batch[j], m1[0], spec.clock_now)
            done += 1
    rec["find_plus_commit_ms_per_pod"] = (time.perf_counter() - t0) * 1e3 / len(batch)
    rec["committed"] = done
    out.append(rec)
    eng.close()
print(json.dumps(out))
