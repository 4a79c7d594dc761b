#!/bin/bash
# Round 3, single-launch find: parity of the new kernel against the staged path, the whole GPU suite on top (HipMatcher's
# FindNode now runs through it), then the latency of the call.
out=gpurun_out/r03_find1; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "single_launch or sparse_candidates" > $out/pytest_new.log 2>&1; echo "new tests rc=$? seconds=$(( $(date +%s) - t0 ))"; tail -5 $out/pytest_new.log
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py::test_single_launch_find_equals_the_staged_path --deselect tests/test_gpu_parity.py::test_mode_b_sparse_candidates_make_windows_run_dry > $out/pytest_full.log 2>&1; echo "suite rc=$? seconds=$(( $(date +%s) - t0 ))"; tail -3 $out/pytest_full.log
timeout 300 python tools/time_findnode.py > $out/findnode_latency.json 2> $out/findnode.err; echo "findnode rc=$?"; cat $out/findnode_latency.json
NHDFIT_LIBRARY=$PWD/nhd_amd/libnhdfit_tuning.so NHDFIT_FIND_PROF=1 timeout 120 python - > $out/find_prof.log 2>&1 <<'PY'
import sys; sys.path.insert(0, '.')
from workload import refmodel, synth
from nhd_amd import pack, planes
from nhd_amd.engine import Engine
spec = synth.make_cluster(4, n_nodes=65536)
pods, groups = synth.make_pods(4, n_pods=16)
tops = [refmodel.make_topology(s) for s in pods]
pk = pack.Packer(); table = planes.planes_from_spec(pk, spec); reqs = pk.digest_many(tops, groups)
eng = Engine(0); eng.set_dictionary(pk); eng.upload(table)
for k in range(12):
    eng.find(reqs[k:k+1], spec.clock_now, want_bitmap=False)
PY
tail -30 $out/find_prof.log
