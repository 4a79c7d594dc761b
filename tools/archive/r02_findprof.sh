#!/bin/bash
# host-side phase times of nhdfit_find for one pod and for 4096 pods (attached matcher / engine)
NHDFIT_FIND_PROF=1 timeout 300 python - <<'PY' 2>&1 | tail -40
import numpy as np, time
from nhd_amd import pack
from nhd_amd.engine import Engine
from workload import planes, refmodel, synth
spec = synth.make_cluster(4, n_nodes=65536)
pods, groups = synth.make_pods(4, n_pods=4096)
tops = [refmodel.make_topology(s) for s in pods]
pk = pack.Packer(); table = planes.planes_from_spec(pk, spec); reqs = pk.digest_many(tops, groups); pk.close_signatures()
eng = Engine(0); eng.set_dictionary(pk); eng.upload(table)
for _ in range(3): eng.find(reqs[:1], spec.clock_now, want_bitmap=False, want_map=True)
print("---- 4096 pods", flush=True)
for _ in range(2): eng.find(reqs, spec.clock_now, want_bitmap=False, want_map=True)
PY
