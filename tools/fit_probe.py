#!/usr/bin/env python3
"""Tuning probe: the bench workload with the mapping roles off (fit + digest only), N pipelined steps.
Run under rocprofv3 with NHDFIT_ROLE_KERNELS=1 to get the stand-alone time of each role."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nhd_amd import pack
from workload import planes, refmodel, synth
from nhd_amd.engine import Engine

cfg = int(os.environ.get("PROBE_CFG", "4"))
n, P, steps = int(os.environ.get("PROBE_N", "65536")), int(os.environ.get("PROBE_P", "4096")), int(os.environ.get("PROBE_STEPS", "100"))
bitmap, mapping = int(os.environ.get("PROBE_BITMAP", "1")), int(os.environ.get("PROBE_MAP", "0"))
spec = synth.make_cluster(cfg, n_nodes=n)
pods, groups = synth.make_pods(cfg, n_pods=P)
tops = [refmodel.make_topology(s) for s in pods]
pk = pack.Packer()
table = planes.planes_from_spec(pk, spec)
reqs = pk.digest_many(tops, groups)
eng = Engine(0)
eng.set_dictionary(pk)
eng.upload(table)
eng.set_outputs(bool(bitmap), bool(mapping))
eng.stage(reqs)
for _ in range(10):
    eng.enqueue(spec.clock_now)
eng.sync(); eng.reset_stats()
t0 = time.perf_counter()
for _ in range(steps):
    eng.enqueue(spec.clock_now)
eng.sync()
dt = time.perf_counter() - t0
st = eng.stats()
print(json.dumps({"us_per_step": dt / steps * 1e6, "kernel_us": st.fit_ms_total / max(1, st.launches) * 1e3, "lds": st.lds_bytes}))
