#!/usr/bin/env python3
"""Generate tests/golden/delta/*.json: release / reclaim / reset and the scheduler's scalar writes on the UNMODIFIED
reference objects (imported from /root/reference).  TEST INFRASTRUCTURE; run in the build container only:

    python oracle/gen_golden_delta.py

Per case (tests/delta_check.py): a seeded synthetic cluster, a pending list scheduled with the reference's own
Matcher.FindNode + SetBusy + SetPhysicalIdsFromMapping + ClaimPodNICResources (nhd/NHDScheduler.py:274-304), then the
deterministic op stream of delta_check.make_ops through the reference's own mutators (Node.AddResourcesFromTopology,
RemoveResourcesFromTopology, ResetResources, SetGroups, SetHugepages, attribute writes) and Matcher.FindNode calls in
between.  The fixture holds the binds, the result of every `find` operation and every node's state in packed terms at
checkpoints - what HipMatcher's device mirror must hold when the same stream runs through attached stand-in objects.
"""
import contextlib
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import nhd_oracle, ref_loader          # noqa: E402
from tests import delta_check as D                 # noqa: E402
from workload import refmodel                      # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "delta")


def main():
    ref = ref_loader.load()
    os.makedirs(OUT, exist_ok=True)
    for cfg, n_nodes, n_pods, n_ops in D.CASES:
        spec, pods, groups = D.workload(cfg, n_nodes, n_pods)
        clock = ref_loader.VirtualClock(spec.clock_now).install()
        nodes = spec.build_nodes(ref)
        tops = [refmodel.make_topology(p, ref) for p in pods]
        binds = []
        for top, grp in zip(tops[:n_pods], groups[:n_pods]):
            res = ref_loader.find_node(nhd_oracle.initial_node_filter(nodes, grp), top)
            binds.append(res[0])
            if res[0] is None:
                continue
            n = nodes[res[0]]
            n.SetBusy()
            with contextlib.redirect_stdout(io.StringIO()):
                nic_list = n.SetPhysicalIdsFromMapping(res[1], top)
            n.ClaimPodNICResources(list({x[0] for x in nic_list}))
        placed = [(i, b) for i, b in enumerate(binds) if b is not None]
        seed = 0xD17A + cfg
        ops = D.make_ops(seed, list(nodes), placed, n_ops, clock.t)
        finds, checkpoints = [], []
        for k, op in enumerate(ops):
            if op[0] == "find":
                j = n_pods + op[2]
                finds.append(D.as_jsonable(ref_loader.find_node(nhd_oracle.initial_node_filter(nodes, groups[j]), tops[j])))
            else:
                with contextlib.redirect_stdout(io.StringIO()):
                    D.apply_op(nodes, tops, op)
            if (k + 1) % D.CHECK_EVERY == 0 or k + 1 == n_ops:
                checkpoints.append({"after": k + 1, "state": D.state_of(nodes)})
        fixture = {"config": cfg, "n_nodes": n_nodes, "n_pods": n_pods, "n_ops": n_ops, "seed": seed, "clock": clock.t,
                   "binds": binds, "finds": finds, "checkpoints": checkpoints}
        path = os.path.join(OUT, f"delta_c{cfg}.json")
        with open(path, "w") as f:
            json.dump(fixture, f, separators=(",", ":"))
        kinds = {}
        for op in ops:
            kinds[op[0]] = kinds.get(op[0], 0) + 1
        print(path, "bound", len(placed), "of", n_pods, kinds, "finds placed", sum(r[0] is not None for r in finds), os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
