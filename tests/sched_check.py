"""Replay of the reference scheduler-loop fixtures (tests/golden/sched, produced by the unmodified NHDScheduler.
AttemptScheduling - oracle/gen_golden_sched.py) through HipMatcher in attached mode: per-pod FindNode + the node mutators,
or one ScheduleBatch for the whole pending list.  Shared by the CPU test (host-twin engine) and the GPU test."""
import glob
import json
import os

import numpy as np

from nhd_amd import pack
from nhd_amd.matcher import HipMatcher
from tests import sched_standin
from workload import refmodel, synth

FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "sched", "*.json")))


def load(path):
    with open(path) as f:
        return json.load(f)


class Clock:
    def __init__(self, t): self.t = t
    def __call__(self): return self.t


def replay(case, batched, engine_factory=None, devices=None):
    spec = synth.make_cluster(case["config"], n_nodes=case["n_nodes"])
    pods, groups = synth.make_pods(case["config"], n_pods=case["n_pods"])
    for p in pods:
        p["misc_smt"] = True
    clock = Clock(case["clock0"])
    nodes = sched_standin.adopt(spec.build_nodes(), clock)
    tops = [refmodel.make_topology(p) for p in pods]
    m = HipMatcher(clock=clock, engine_factory=engine_factory, devices=devices)
    uploads = []
    orig_upload = m.engine.upload
    m.engine.upload = lambda *a, **k: (uploads.append(1), orig_upload(*a, **k))[1]
    m.attach(nodes)
    n_attach_uploads = len(uploads)
    import time
    t_loop = time.perf_counter()
    if batched:
        # the reference advances its clock per pod; a batch is matched at one instant: only valid when no busy window
        # (30 s) can expire inside the batch - true for the fixtures (dt * pods < 30 s is NOT required: SetBusy times only
        # move forward and `now` is the last pod's time would differ) - so the batched replay uses dt = 0 fixtures only
        binds = sched_standin.check_pending_pods_batched(nodes, m, tops, groups, now=clock.t)
    else:
        def tick():
            clock.t += case["dt"]
        binds = sched_standin.check_pending_pods(nodes, m, tops, groups, tick=tick)
    m.loop_seconds = time.perf_counter() - t_loop
    return nodes, m, binds, len(uploads) - n_attach_uploads


def packed(nodes):
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


def mirror_state(m):
    """The device mirror, downloaded, in the same packed terms."""
    t = m.engine.download()
    out = {}
    for i, name in enumerate(m._names):
        d = t.detail[i]
        out[name] = {"t0": [int(x) for x in t.p0[i]["t0"]], "t1": [int(x) for x in t.p1[i]["t1"]],
                     "gpu_free": int(t.p2[i]["gpu_free"]), "hp_free": int(t.p2[i]["hp_free"]), "busy_time": float(t.p4[i]["busy_time"]),
                     "nic_claimed": [[int(d["nic_cls"][u][k]) == 0 for k in range(int(d["nic_cnt"][u]))] for u in range(2)],
                     "sw_free": [int(x) for x in d["sw_free"]]}
    return out


def check_per_pod(case, engine_factory=None, devices=None):
    nodes, m, binds, extra_uploads = replay(case, batched=False, engine_factory=engine_factory, devices=devices)
    assert binds == case["binds"]
    assert sum(b is None for b in binds) == case["failed_schedule_count"]
    assert packed(nodes) == case["final"]                     # the node objects went the reference's way ...
    assert mirror_state(m) == case["final"]                   # ... and so did the device mirror,
    assert extra_uploads == 0                                  # without a single node being re-packed and re-uploaded
    return sum(b is not None for b in binds)
