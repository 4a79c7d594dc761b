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
The four BASELINE-shaped cases keep misc_cores_smt enabled.  `commit_q1_c2.json` draws the pods as the generator does
(misc_cores_smt on or off) on a cluster small enough to fill up: FilterNumaTopology halves the misc-core demand whatever that
flag says (quirk Q1, nhd/Matcher.py:178-204), so a socket can win with fewer free physical cores than the pod's misc cores, and
GetFreeCpuBatch (nhd/Node.py:502-519) then walks on into the sibling range and hands out second threads as cores of their own.
The sequence ends before the first commit the reference raises on (its unwind path is broken, SURVEY.md Appendix B: parity is
undefined from there); the fixture says how many of its pods met the run-on walk.
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
CASES = [(3, 40, 120, True, "commit_c3"), (4, 24, 150, True, "commit_c4"), (5, 60, 200, True, "commit_c5"), (2, 16, 60, True, "commit_c2"),
         (2, 48, 160, False, "commit_q1_c2")]          # (config, nodes, pods, misc_cores_smt forced on, file)


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
    for cfg, n_nodes, n_pods, force_smt, fname in CASES:
        spec = synth.make_cluster(cfg, n_nodes=n_nodes)
        clock = ref_loader.VirtualClock(spec.clock_now).install()
        pods, groups = synth.make_pods(cfg, n_pods=n_pods)
        if force_smt:
            for p in pods:
                p["misc_smt"] = True
        nodes = spec.build_nodes(ref)
        expected = []
        run_on = 0
        for p, grp in zip(pods, groups):
            top = refmodel.make_topology(p, ref)
            sub = nhd_oracle.initial_node_filter(nodes, grp)
            res = ref_loader.find_node(sub, top)
            if res[0] is None:
                expected.append([None])
                continue
            n = nodes[res[0]]
            n.SetBusy()
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    nic_list = n.SetPhysicalIdsFromMapping(res[1], top)
            except IndexError:                                 # the reference's own failure: the defined sequence ends before this pod
                assert not force_smt, "a BASELINE-shaped case met a commit the reference raises on"
                # (the raising call already wrote into the node: the final state is the one before this pod's commit - take it from
                #  a replay of the defined prefix below)
                break
            n.ClaimPodNICResources(list({x[0] for x in nic_list}))
            m = res[1]
            ids = ids_of(top, n)
            num_cores = int(n.cores_per_proc) * int(n.sockets)
            lists = [(ids["misc"], p["misc_smt"])] + [(g["cores"], pg["proc_smt"]) for g, pg in zip(ids["groups"], p["groups"])] + \
                    [(g["helpers"], pg["helper_smt"]) for g, pg in zip(ids["groups"], p["groups"])]
            if any(not smt and any(c >= num_cores for c in lst) for lst, smt in lists):
                run_on += 1                                    # a request for whole cores was handed a second thread as a core of its own
            expected.append([res[0], {"gpu": list(m["gpu"]), "cpu": list(m["cpu"]), "nic": [list(x) for x in m["nic"]]}, ids])
        if len(expected) < n_pods:                             # replay the defined prefix on fresh objects for the final state
            n_def = len(expected)
            clock = ref_loader.VirtualClock(spec.clock_now).install()
            nodes = spec.build_nodes(ref)
            for p, grp, want in zip(pods[:n_def], groups[:n_def], expected):
                top = refmodel.make_topology(p, ref)
                res = ref_loader.find_node(nhd_oracle.initial_node_filter(nodes, grp), top)
                assert (res[0] is None) == (want[0] is None) and (res[0] is None or res[0] == want[0])
                if res[0] is None:
                    continue
                n = nodes[res[0]]
                n.SetBusy()
                with contextlib.redirect_stdout(io.StringIO()):
                    nic_list = n.SetPhysicalIdsFromMapping(res[1], top)
                n.ClaimPodNICResources(list({x[0] for x in nic_list}))
        fixture = {"config": cfg, "n_nodes": n_nodes, "n_pods": len(expected), "clock": clock.t, "expected": expected, "final": packed_state(nodes)}
        if len(expected) != n_pods:
            fixture["n_pods_drawn"] = n_pods                   # (the generator's pods depend on how many are drawn: draw these, keep the first n_pods)
        if not force_smt:
            fixture["force_misc_smt"] = False
            fixture["run_on_pods"] = run_on
            assert run_on > 0, "the case is meant to meet GetFreeCpuBatch's walk into the sibling range"
        path = os.path.join(OUT, fname + ".json")
        with open(path, "w") as f:
            json.dump(fixture, f, separators=(",", ":"))
        print(path, "placed", sum(e[0] is not None for e in expected), "of", len(expected), "defined pods;", run_on, "met the run-on walk")


if __name__ == "__main__":
    main()
