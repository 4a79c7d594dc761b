"""Offline soak (build container only: needs /root/reference; `python tools/soak_big_reference.py <seeds>`): pods with 5..7 processing
groups through the UNMODIFIED reference Matcher, the Python oracle and the product's host build (HipMatcher on tests/harness) on
random clusters of ordinary and wide nodes - the three must agree on node and mapping."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from nhd_amd.matcher import HipMatcher  # noqa: E402
from oracle import nhd_oracle as O, ref_loader  # noqa: E402
from tests import harness, util  # noqa: E402
from tests.test_big_core import big_spec  # noqa: E402
from tests.test_big_vs_reference import few_nics  # noqa: E402
from tests.test_wide_core import norm  # noqa: E402
from workload import refmodel  # noqa: E402

ref = ref_loader.load()
ref_loader.VirtualClock(util.CLOCK).install()
t0 = time.time()
bad = placed = pods = 0
for seed in range(int(sys.argv[1])):
    wide = 0.35 if seed % 2 else 0.0
    descs = few_nics(util.mixed_cluster_desc(7100 + seed, 12, wide_share=wide, occupancy=0.08), 3)
    nl_ref, nl = util.build_cluster(descs, ref), util.build_cluster(descs)
    rng = np.random.default_rng(1900 + seed)
    specs = [big_spec(rng, 5, 6 if wide else 7) for _ in range(6)]
    m = HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine)
    got = m.FindNodes(nl, [refmodel.make_topology(s) for s in specs])
    for s, g in zip(specs, got):
        want = ref_loader.find_node(nl_ref, refmodel.make_topology(s, ref))
        pods += 1
        placed += want[0] is not None
        if O.find_node(nl, refmodel.make_topology(s), util.CLOCK) != want:
            bad += 1
            print("ORACLE != REFERENCE seed", seed, s, flush=True)
        if norm(g) != norm(want):
            bad += 1
            print("PRODUCT != REFERENCE seed", seed, s, norm(g), norm(want), flush=True)
    if seed % 10 == 9:
        print("seed", seed, "pods", pods, "placed", placed, "mismatches", bad, "seconds", round(time.time() - t0, 1), flush=True)
print("seeds", sys.argv[1], "pods", pods, "placed by the reference", placed, "mismatches", bad, "seconds", round(time.time() - t0, 1))
