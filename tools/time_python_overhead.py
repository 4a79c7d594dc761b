#!/usr/bin/env python3
"""CPU only: what HipMatcher.FindNode costs in Python around the device call, measured with an engine stub that answers at once
(attached mirror of 4 096 nodes; the scheduler's own dict, and a filtered dict as InitialNodeFilter hands it over).  `--profile`
prints the top functions."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from nhd_amd import pack  # noqa: E402
from nhd_amd.matcher import HipMatcher  # noqa: E402
from tests import harness  # noqa: E402
from workload import refmodel, synth  # noqa: E402


class Stub(harness.HarnessEngine):
    """an instant device: a canned winner for every pod"""

    def find(self, reqs, now, cand=None, want_bitmap=True, want_map=True):
        P = len(reqs)
        score = np.full(P, (1 << 63) | (0x7FFFFFFFFFFFFFFF - 5), np.uint64)
        maps = np.zeros(P, pack.MAPPING)
        maps["valid"] = 1
        return score, None, maps


spec = synth.make_cluster(4, n_nodes=4096)
nl = spec.build_nodes()
pods, groups = synth.make_pods(4, n_pods=200)
tops = [refmodel.make_topology(s) for s in pods]
m = HipMatcher(clock=lambda: spec.clock_now, engine_factory=Stub)
m.attach(nl)


def per_call(d, rounds=5):
    for t in tops[:20]:
        m.FindNode(d, t)
    t0 = time.perf_counter()
    for _ in range(rounds):
        for t in tops:
            m.FindNode(d, t)
    return (time.perf_counter() - t0) / (rounds * len(tops)) * 1e6


print("FindNode(the attached dict): %.1f us of Python per call" % per_call(nl))
sub = {k: v for i, (k, v) in enumerate(nl.items()) if i % 3}
print("FindNode(a filtered dict of %d nodes, the same one call after call): %.1f us" % (len(sub), per_call(sub)))
if "--profile" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    per_call(nl, rounds=3)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(12)
