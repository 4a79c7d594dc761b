#!/usr/bin/env python3
"""Generate tests/golden/commit/*.json: the UNMODIFIED reference's commit step (imported from /root/reference).
TEST INFRASTRUCTURE; run in the build container only:

    python oracle/gen_golden_commit.py

For a seeded synthetic cluster and pod list (workload.synth), the scheduler loop of nhd/NHDScheduler.py:274-304 is
replayed with the reference's own objects under a virtual clock: InitialNodeFilter -> Matcher.FindNode -> SetBusy ->
SetPhysicalIdsFromMapping -> ClaimPodNICResources, pod after pod.  Each fixture holds
  expected[i] = [node name, mapping, ids]   ids = the physical ids the reference wrote into the pod's CfgTopology:
                {'groups': [{'cores': batch (GPU cpu_cores first, then proc_cores), 'helpers': [...],
                             'gpus': positions in Node.gpus}], 'misc': [...]}          or [None]
  final[name] = the node's state afterwards in packed terms: thread-0 / thread-1 free-core masks per socket, free-GPU
                mask, free hugepages, busy time, which NICs are claimed (per NUMA node, by ordinal), free GPUs per
                local PCIe switch id - what the device mirror must hold after nhdfit_schedule_batch(apply).
Pods keep misc_cores_smt enabled: the reference's own unwind path is broken, parity is
undefined where a commit fails.
"""
import contextlib
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from nhd_amd import pack
from workload import refmodel, synth# noqa: E402
from oracle import nhd_oracle, ref_loader    # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "commit")
CASES = [(3, 40, 120), (4, 24, 150), (5, 60, 200), (2, 16, 60)]


def ids_of(top, node):
    pos = {g.device_id: i for i, g in enumerate(node.gpus)}
    groups = []
    for pg in top.proc_groups:
        groups.append({"cores": [c.core for g in pg.group_gpus for c in g.cpu_cores] + [c.core for c in pg.proc_cores],
                       "helpers": [c.core for c in pg.misc_cores],
                       "gpus": [pos[g.device_id] for g in pg.group_gpus]})
    return {"groups": groups, "misc": [c.core for c in top.misc_cores]}


def packed_state(nodes):
    pk = pack.Packer()
    t = pk.pack_nodes(nodes)
    out = {}
    for i, name in enumerate(t.names):
        d = t.detail[i]
        out[name] = {"t0": [int(x) for x in t.p0[i]["t0"]], "t1": [int(x) for x in t.p1[i]["t1"]],
                     "gpu_free": int(t.p2[i]["gpu_free"]), "hp_free": int(t.p2[i]["hp_free"]), "busy_time": float(t.p4[i]["busy_time"]),
                     "nic_claimed": [[int(d["nic_cls"][u][k]) == 0 for k in range(int(d["nic_cnt"][u]))] for u in range(2)],
                     "sw_free": [int(x) for x in d["sw_free"]]}
    return out


def main():
    ref = ref_loader.load()
    os.makedirs(OUT, exist_ok=True)
    for cfg, n_nodes, n_pods in CASES:
        spec = synth.make_cluster(cfg, n_nodes=n_nodes)
        clock = ref_loader.VirtualClock(spec.clock_now).install()
        pods, groups = synth.make_pods(cfg, n_pods=n_pods)
        for p in pods:
            p["misc_smt"] = True
        nodes = spec.build_nodes(ref)
        expected = []
        for p, grp in zip(pods, groups):
            top = refmodel.make_topology(p, ref)
            sub = nhd_oracle.initial_node_filter(nodes, grp)
            res = ref_loader.find_node(sub, top)
            if res[0] is None:
                expected.append([None])
                continue
            n = nodes[res[0]]
            n.SetBusy()
            with contextlib.redirect_stdout(io.StringIO()):
                nic_list = n.SetPhysicalIdsFromMapping(res[1], top)
            n.ClaimPodNICResources(list({x[0] for x in nic_list}))
            m = res[1]
            expected.append([res[0], {"gpu": list(m["gpu"]), "cpu": list(m["cpu"]), "nic": [list(x) for x in m["nic"]]}, ids_of(top, n)])
        fixture = {"config": cfg, "n_nodes": n_nodes, "n_pods": n_pods, "clock": clock.t, "expected": expected, "final": packed_state(nodes)}
        path = os.path.join(OUT, f"commit_c{cfg}.json")
        with open(path, "w") as f:
            json.dump(fixture, f, separators=(",", ":"))
        print(path, "placed", sum(e[0] is not None for e in expected), "of", n_pods)


if __name__ == "__main__":
    main()
