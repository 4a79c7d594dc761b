"""Offline soak (not part of the suite; `python tools/soak_big.py <seeds>`, CPU only) of the general path for requests (pods with 5..8
processing groups, nhdfit_big_req) on the host build: (1) ordinary pods digested both ways - the general path's verdict for every
(pod, node) pair, score word and winner's mapping against the table-driven pass; (2) clusters of ordinary and wide nodes with pods of up
to eight groups - FindNodes and ScheduleBatch (decisions, mappings, physical ids) against the Python oracle."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from nhd_amd import pack  # noqa: E402
from nhd_amd.matcher import HipMatcher  # noqa: E402
from oracle import nhd_oracle as O  # noqa: E402
from tests import harness, util  # noqa: E402
from tests.test_big_core import big_spec  # noqa: E402
from tests.test_wide_core import norm, unpack  # noqa: E402
from workload import refmodel  # noqa: E402


def few_nics(descs, most=4):
    for d in descs:
        keep, lab = 0, {}
        for k, v in d["labels"].items():
            if "nfd-extras-nic" in k:
                keep += 1
                if keep > most:
                    continue
            lab[k] = v
        d["labels"] = lab
        d["nic_pods_used"] = d["nic_pods_used"][:sum(1 for k in lab if "nfd-extras-nic" in k and "10000Mbs" not in k.replace("100000Mbs", ""))]
    return descs


t0 = time.time()
bad = 0
n = int(sys.argv[1])
EF = {} if "--device" in sys.argv else {"engine_factory": harness.HarnessEngine}      # --device: the same through the C-ABI on the GPU
for seed in range(n):
    # (1) two evaluations of one predicate
    nl = util.mixed_cluster(91000 + seed, 48, wide_share=0.25 if seed % 2 else 0.0)
    rng = np.random.default_rng(seed)
    tops = [refmodel.make_topology(util.random_pod_spec(rng, max_groups=4)) for _ in range(40)]
    m = HipMatcher(clock=lambda: util.CLOCK, **EF)
    m.FindNodes(nl, tops[:1])
    reqs = m.packer.digest_many(tops)
    big = np.array([m.packer.digest_big(t) for t in tops], dtype=pack.BIG_REQ)
    score, bm, maps = m.engine.find(reqs, util.CLOCK, want_bitmap=True, want_map=True)
    if EF:
        fits, bscore, exhausted = harness.big_eval(m.packer, m.engine.table, m.engine._wide_records(), big, util.CLOCK)
        _, bmaps = m.engine.big_find(big, util.CLOCK)
        ok = not exhausted and np.array_equal(unpack(bm, len(nl)).astype(np.uint8), fits.T) and np.array_equal(score, bscore)
    else:                                      # on the device: the general path's score words and mappings against the table pass's
        bscore, bmaps = m.engine.big_find(big, util.CLOCK)
        ok = np.array_equal(score, bscore)
    for p in np.flatnonzero(score != 0):
        G = int(reqs[p]["n_groups"])
        ok = ok and all(list(maps[p][f][:k]) == list(bmaps[p][f][:k]) for f, k in (("gpu", G), ("cpu", G + 1), ("nic_numa", G), ("nic_idx", G)))
    if not ok:
        bad += 1
        print("TWO-WAY MISMATCH seed", seed, flush=True)
    # (2) big pods against the oracle
    if seed % 3 == 0:
        descs = few_nics(util.mixed_cluster_desc(93000 + seed, 24, wide_share=0.3 if seed % 2 else 0.0, occupancy=0.1))
        nl, ref_nl = util.build_cluster(descs), util.build_cluster(descs)
        specs = []
        for _ in range(22):
            s = big_spec(rng, 5, 7) if rng.random() < 0.6 else util.random_pod_spec(rng)
            s["misc_smt"] = True
            if s["map_type"] == "NONE":
                s["map_type"] = "NUMA"
            specs.append(s)
        tops = [refmodel.make_topology(s) for s in specs]
        m = HipMatcher(clock=lambda: util.CLOCK, **EF)
        if [norm(r) for r in m.FindNodes(nl, tops)] != [norm(O.find_node(nl, t, util.CLOCK)) for t in tops]:
            bad += 1
            print("FIND MISMATCH seed", seed, flush=True)
        m.attach(nl)
        got = m.ScheduleBatch(nl, tops, now=util.CLOCK)
        want, ids = [], []
        for top in tops:
            r = O.find_node(ref_nl, top, util.CLOCK)
            rec = {}
            if r[0] is not None:
                try:
                    O.commit(ref_nl[r[0]], top, r[1], util.CLOCK, rec)
                except O.CommitFailure:
                    break
            want.append(norm(r))
            ids.append(rec if r[0] is not None else None)
        k = len(want)
        if [norm(r) for r in got[:k]] != want or m.last_placements[:k] != ids:
            bad += 1
            print("MODE B MISMATCH seed", seed, flush=True)
    if seed % 20 == 19:
        print("seed", seed, "mismatches so far", bad, "seconds", round(time.time() - t0, 1), flush=True)
print("seeds", n, "mismatches", bad, "seconds", round(time.time() - t0, 1))
