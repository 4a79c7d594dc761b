#!/bin/bash
# Round 4, the GPU call after the general path for requests (nhdfit_big_req, ABI 8) went in: the whole GPU suite (no -x: every
# test runs, the new ones are the last of test_gpu_parity.py), smoke(), the driver's bench form with its extras (big_pod_find).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_big
mkdir -p $OUT
cd $ROOT
SECONDS=0
timeout 330 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$? seconds=$SECONDS" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_gpu.log | tail -12
timeout 120 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
echo "seconds=$SECONDS"
timeout 200 python bench.py --steps 20 --warmup 5 --no-pmc > $OUT/bench_driver_form.json 2> $OUT/bench.err
echo "bench rc=$? seconds=$SECONDS"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04_big/bench_driver_form.json").readline())
    print("value", d["value"], "ms_per_step", d["ms_per_step"], "big_pod_find", d.get("big_pod_find"), "mode_b", d.get("mode_b", {}).get("decisions_per_s"))
except Exception as e:
    print("no bench line:", e)
PY
