#!/usr/bin/env python3
"""Soak of mode B's decision engine on the device (`python tools/soak_mode_b_gpu.py <seeds> [first]` on a GPU box): BASELINE configs 2..5
drawn with other seeds at random sizes (500..24 000 nodes, 100..3 000 pods; every fourth seed a SMALL cluster of 20..400 nodes that fills
up long before the pods run out; every fifth seed with nine nodes in ten under maintenance, so that the decision engine's windows run dry;
every third seed decides the batch once without applying it first - the undo path - and then for good) through nhdfit_schedule_batch(apply) against the independent
oracle (oracle/seq_oracle.py: C scan + C commit, Python set-order mapping) - node, mapping and physical ids of every pod, the mirror of
every node that received one (tests/test_gpu_parity.mode_b_against_seq_oracle).  A batch in which the oracle meets a commit the
reference raises on is cut in front of that pod."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nhd_amd import pack
from nhd_amd.engine import Engine
from oracle import coracle, seq_oracle
from tests.test_gpu_parity import mode_b_against_seq_oracle
from workload import planes, refmodel, synth

n_seeds = int(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t0 = time.time(); bad = pods_total = placed_total = cut = 0
for seed in range(first, first + n_seeds):
    rng = np.random.default_rng(424200 + seed)
    cfg = int(rng.choice([2, 3, 4, 5]))
    n = int(rng.integers(500, 24000))
    P = int(rng.integers(100, 3000))
    if seed % 4 == 1:
        n = int(rng.integers(20, 400))
    spec = synth.make_cluster(cfg, n_nodes=n, seed=7000 + seed)
    if seed % 5 == 2:
        keep = rng.random(n) < 0.1
        spec.maintenance[:] = ~keep
    pods, groups = synth.make_pods(cfg, n_pods=P, seed=7000 + seed)
    tops = [refmodel.make_topology(s) for s in pods]
    # where would the reference raise?  (the oracle alone, on its own copy)
    sc = seq_oracle.SeqCluster(coracle.Cluster.from_spec(spec))
    _, _, _, n_def = seq_oracle.schedule_sequence(sc, tops, groups, spec.clock_now)
    if n_def < len(tops):
        cut += 1
        pods, groups, tops = pods[:n_def], groups[:n_def], tops[:n_def]
    if not tops:
        continue
    pk = pack.Packer()
    table = planes.planes_from_spec(pk, spec)
    reqs = pk.digest_many(tops, groups)
    pk.close_signatures()
    eng = Engine(0)
    eng.set_dictionary(pk)
    eng.upload(table)
    try:
        if seed % 3 == 0:                                    # decided, undone (apply=False), then decided again below: the mirror must be back where it was
            dry = eng.schedule_batch(reqs, spec.clock_now, pk, apply=False)
        placed, distinct = mode_b_against_seq_oracle(spec, pods, groups, eng, pk, reqs, tops, spec.clock_now)
        pods_total += len(tops); placed_total += placed
        if seed % 3 == 0:
            assert placed == int((np.asarray(dry[0]) >= 0).sum()), "the dry run placed another number of pods"
    except AssertionError as e:
        bad += 1
        print("MISMATCH seed", seed, "config", cfg, "nodes", n, "pods", len(tops), str(e)[:300], flush=True)
    eng.close()
    if (seed - first) % 10 == 9:
        print("seed", seed, "pods", pods_total, "placed", placed_total, "cut short", cut, "mismatches", bad, "seconds", round(time.time() - t0, 1), flush=True)
print("seeds", n_seeds, "pods", pods_total, "placed", placed_total, "batches cut short", cut, "mismatches", bad, "seconds", round(time.time() - t0, 1))
sys.exit(1 if bad else 0)
