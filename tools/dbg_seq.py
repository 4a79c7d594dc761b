#!/usr/bin/env python3
"""Debug aid: device mode B without the signature closure vs the host twin with it; prints the first differences."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nhd_amd import pack
from nhd_amd.engine import Engine
from workload import refmodel, synth
from tests import harness

cfg, n, P = 4, 48, 200
spec = synth.make_cluster(cfg, n_nodes=n)
pods, groups = synth.make_pods(cfg, n_pods=P)
for p in pods:
    p["misc_smt"] = True
tops = [refmodel.make_topology(s) for s in pods]
pk2 = pack.Packer(); t2 = pk2.pack_nodes(spec.build_nodes()); r2 = pk2.digest_many(tops, groups); pk2.close_signatures()
hn, hm, hp, hs, done = harness.schedule(pk2, t2, r2, spec.clock_now, apply=True)
for trial in range(4):
    pk = pack.Packer(); table = pk.pack_nodes(spec.build_nodes()); reqs = pk.digest_many(tops, groups)
    if trial >= 2:
        pk.close_signatures()
    eng = Engine(0); eng.set_dictionary(pk); eng.upload(table)
    node, maps, places, status = eng.schedule_batch(reqs, spec.clock_now, pk, apply=True)
    bad = np.nonzero(node != hn)[0]
    need = reqs["gpus"].sum(axis=1) > 0
    print("trial", trial, "closure" if trial >= 2 else "no closure", "mismatches", len(bad), "new_sig pods", np.nonzero(status == 2)[0].tolist())
    for i in bad[:6]:
        print("  pod", i, "gpu" if need[i] else "nogpu", "device", node[i], "twin", hn[i], "status", status[i], hs[i])
    eng.close()
