"""Offline soak (not part of the suite; `python tools/soak_cpu.py <seeds> [<delta streams>]`, CPU only): many more seeds of the CPU differential tests - host twin of the kernel arithmetic
vs the Python oracle: find (bitmap, winner, mapping; the lone-pod form against the table form), mode B with commits + physical ids, deltas vs stand-in mutators."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nhd_amd import pack
from tests import util, harness, sched_standin
from tests import delta_check as D
from workload import refmodel
from oracle import nhd_oracle as O
from tests.test_core_vs_oracle import decode, bitmap_rows

t0 = time.time()
bad = 0
for seed in range(1000, 1000 + int(sys.argv[1])):
    nl = util.random_cluster(31000 + seed, 40)
    rng = np.random.default_rng(seed)
    specs = [util.random_pod_spec(rng, max_groups=4 if seed % 4 == 0 else 3) for _ in range(30)]
    tops = [refmodel.make_topology(s) for s in specs]
    pk = pack.Packer(); table = pk.pack_nodes(nl); reqs = pk.digest_many(tops)
    score, bitmap, maps = harness.find(pk, table, reqs, util.CLOCK)
    ls, lb, _ = harness.find_lone(pk, table, reqs, util.CLOCK)           # the lone-pod form (k_find1's arithmetic): the same scores and bitmaps
    if not (np.array_equal(ls, score) and np.array_equal(lb, bitmap)):
        bad += 1; print("LONE FORM MISMATCH seed", seed)
    got = decode(pk, table, reqs, score, maps, table.names)
    rows = bitmap_rows(bitmap, table.n, len(reqs))
    for p, top in enumerate(tops):
        want = O.find_node(nl, top, util.CLOCK)
        want = [None] if want[0] is None else [want[0], {"gpu": list(want[1]["gpu"]), "cpu": list(want[1]["cpu"]), "nic": [list(x) for x in want[1]["nic"]]}]
        if got[p] != want or rows[p] != "".join("1" if O.feasible(v, top, util.CLOCK) else "0" for v in nl.values()):
            bad += 1; print("FIND MISMATCH seed", seed, "pod", p, specs[p], got[p], want)
    # mode B with commits
    nl2 = util.random_cluster(71000 + seed, 30, occupancy=0.15); ref_nl = util.random_cluster(71000 + seed, 30, occupancy=0.15)
    specs = []
    for _ in range(60):
        s = util.random_pod_spec(rng); s["misc_smt"] = True
        if s["map_type"] == "NONE": s["map_type"] = "NUMA"
        specs.append(s)
    tops = [refmodel.make_topology(s) for s in specs]
    from nhd_amd.matcher import HipMatcher
    m = HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine); m.attach(nl2)
    try:
        res = m.ScheduleBatch(nl2, tops, now=util.CLOCK)
    except AssertionError:
        continue
    want, ids = [], []
    for top in tops:
        r = O.find_node(ref_nl, top, util.CLOCK); rec = {}
        if r[0] is not None:
            try: O.commit(ref_nl[r[0]], top, r[1], util.CLOCK, rec)
            except O.CommitFailure: break
        want.append(r); ids.append(rec if r[0] is not None else None)
    k = len(want)
    if [D.as_jsonable(x) for x in res[:k]] != [D.as_jsonable(w) for w in want] or m.last_placements[:k] != ids:
        bad += 1; print("MODE B MISMATCH seed", seed)
print("seeds", sys.argv[1], "mismatches", bad, "seconds", round(time.time() - t0, 1))

# release / reclaim / reset / scalar-write streams through the attached matcher: objects == mirror at the end
from workload import synth
n_streams = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t0 = time.time(); bad = applied = repacked = 0
for s in range(n_streams):
    cfg = (3, 4, 5, 2)[s % 4]
    spec = synth.make_cluster(cfg, n_nodes=24)
    case = {"config": cfg, "n_nodes": 24, "n_pods": 50, "n_ops": 300, "seed": 9000 + s, "clock": spec.clock_now}
    m, nodes, binds, finds, uploads = D.replay(case, engine_factory=harness.HarnessEngine, check=False)
    m.FindNode(nodes, refmodel.make_topology(synth.make_pods(cfg, n_pods=1)[0][0]))
    if D.state_of(nodes) != D.mirror_state(m):
        bad += 1; print("DELTA MISMATCH stream", s, cfg)
    applied += m.delta_stats["applied"]; repacked += m.delta_stats["repacked"]
if n_streams:
    print("delta streams", n_streams, "mismatches", bad, "deltas", applied, "re-packed", repacked, "seconds", round(time.time() - t0, 1))
