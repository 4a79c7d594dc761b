#!/usr/bin/env python3
"""Soak of the delta path on the device (`python tools/soak_deltas_gpu.py <streams>` on a GPU box): release / reclaim / reset / scalar-write
streams through the attached matcher (tests/delta_check.replay with the device engine) - node objects == device mirror at the end, and
a FindNode behind every stream.  The host twin's form of the same: tools/soak_cpu.py <seeds> <streams>."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import delta_check as D
from workload import refmodel, synth

n_streams = int(sys.argv[1])
t0 = time.time(); bad = applied = repacked = 0
for s in range(n_streams):
    cfg = (3, 4, 5, 2)[s % 4]
    spec = synth.make_cluster(cfg, n_nodes=24 + 8 * (s % 5))
    case = {"config": cfg, "n_nodes": 24 + 8 * (s % 5), "n_pods": 50, "n_ops": 300, "seed": 19000 + s, "clock": spec.clock_now}
    m, nodes, binds, finds, uploads = D.replay(case, engine_factory=None, check=False)
    m.FindNode(nodes, refmodel.make_topology(synth.make_pods(cfg, n_pods=1)[0][0]))
    if D.state_of(nodes) != D.mirror_state(m):
        bad += 1; print("DELTA MISMATCH stream", s, cfg, flush=True)
    applied += m.delta_stats["applied"]; repacked += m.delta_stats["repacked"]
    m.engine.close()
print("delta streams", n_streams, "mismatches", bad, "deltas", applied, "re-packed", repacked, "seconds", round(time.time() - t0, 1))
sys.exit(1 if bad else 0)
