#!/usr/bin/env python3
"""Generate tests/golden/sched/*.json: the UNMODIFIED reference scheduler loop.  TEST INFRASTRUCTURE; build container only:

    python oracle/gen_golden_sched.py

`NHDScheduler.AttemptScheduling` (nhd/NHDScheduler.py:249-353) is called once per pending pod, in order, exactly as
`CheckPendingPods` (:425-437) does, with the reference's own Matcher, Node objects and commit step, a fake K8s behind it
(oracle/sched_stub.py) and a virtual clock that advances `dt` seconds per pod.  Each fixture holds, for a seeded synthetic
cluster and pod list (workload.synth):
  results[i]  True / False as AttemptScheduling returned it,   binds[i]  the node the pod was bound to or null
  failed_schedule_count, and final[name] = every node's state afterwards in packed terms (as tests/golden/commit).
"""
import contextlib
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_loader, sched_stub                    # noqa: E402
from oracle.gen_golden_commit import packed_state            # noqa: E402
from workload import refmodel, synth                         # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "sched")
CASES = [(3, 40, 100, 1.0), (4, 32, 120, 0.5), (5, 64, 160, 1.0)]


def main():
    ref = ref_loader.load()
    S = sched_stub.load_scheduler_module()
    os.makedirs(OUT, exist_ok=True)
    for cfg, n_nodes, n_pods, dt in CASES:
        spec = synth.make_cluster(cfg, n_nodes=n_nodes)
        clock = ref_loader.VirtualClock(spec.clock_now).install()
        pods, groups = synth.make_pods(cfg, n_pods=n_pods)
        for p in pods:
            p["misc_smt"] = True
        names = [f"pod{i}" for i in range(n_pods)]
        sched, fake = sched_stub.make_scheduler(S, spec.build_nodes(ref), dict(zip(names, groups)))
        results = []
        for name, p in zip(names, pods):
            clock.t += dt
            top = refmodel.make_topology(p, ref)
            sched.GetCfgParser = lambda t, s, _t=top: sched_stub.FakeParser(_t)
            with contextlib.redirect_stdout(io.StringIO()):
                results.append(bool(sched.AttemptScheduling(name, "ns")))
        fixture = {"config": cfg, "n_nodes": n_nodes, "n_pods": n_pods, "clock0": spec.clock_now, "dt": dt, "results": results,
                   "binds": [fake.binds.get(nm) for nm in names], "failed_schedule_count": sched.failed_schedule_count,
                   "final": packed_state(sched.nodes)}
        path = os.path.join(OUT, f"sched_c{cfg}.json")
        with open(path, "w") as f:
            json.dump(fixture, f, separators=(",", ":"))
        print(path, "bound", sum(results), "of", n_pods)


if __name__ == "__main__":
    main()
