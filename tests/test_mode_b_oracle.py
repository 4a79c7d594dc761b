"""Pins the oracle's commit step / sequential batch (mode B) to the unmodified reference: the reference's own
SetBusy + SetPhysicalIdsFromMapping + ClaimPodNICResources applied between FindNode calls."""
import contextlib
import io

import numpy as np
import pytest

from workload import refmodel, synth
from oracle import nhd_oracle as O


def node_state(n):
    return ([c.used for c in n.cores], [g.used for g in n.gpus], [(k.pods_used, tuple(k.speed_used)) for k in n.nics],
            n.mem.free_hugepages_gb, n.busy_time)


@pytest.mark.parametrize("cfg", [2, 3, 4, 5])
def test_sequential_commit_matches_reference(ref, cfg):
    from oracle import ref_loader
    clock = ref_loader.VirtualClock(1.0e6).install()
    spec = synth.make_cluster(cfg, n_nodes=30)
    pods, groups = synth.make_pods(cfg, n_pods=80)
    for p in pods:
        p["misc_smt"] = True                  # stay out of the reference's buggy unwind path
    ref_nodes = spec.build_nodes(ref)
    ora_nodes = spec.build_nodes()            # stand-ins
    tops_r = [refmodel.make_topology(p, ref) for p in pods]
    tops_o = [refmodel.make_topology(p) for p in pods]
    want, want_ids = [], []
    for top, grp in zip(tops_r, groups):
        sub = O.initial_node_filter(ref_nodes, grp)
        res = ref_loader.find_node(sub, top)
        want.append(res)
        ids = None
        if res[0] is not None:
            n = ref_nodes[res[0]]
            n.SetBusy()
            with contextlib.redirect_stdout(io.StringIO()):
                nic_list = n.SetPhysicalIdsFromMapping(res[1], top)
            n.ClaimPodNICResources(list({x[0] for x in nic_list}))
            pos = {g.device_id: i for i, g in enumerate(n.gpus)}        # what the reference wrote into the pod's topology
            ids = {"groups": [{"cores": [c.core for g in pg.group_gpus for c in g.cpu_cores] + [c.core for c in pg.proc_cores],
                               "helpers": [c.core for c in pg.misc_cores], "gpus": [pos[g.device_id] for g in pg.group_gpus]}
                              for pg in top.proc_groups], "misc": [c.core for c in top.misc_cores]}
        want_ids.append(ids)
    got_ids = []
    got = O.schedule_sequence(ora_nodes, tops_o, groups, clock.t, ids_out=got_ids)
    assert got == want
    assert got_ids == want_ids
    assert sum(r[0] is not None for r in want) >= 10
    for k in ref_nodes:
        assert node_state(ref_nodes[k]) == node_state(ora_nodes[k]), k
