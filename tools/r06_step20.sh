#!/bin/bash
# Round 6, GPU call: is the headline region (one region right behind `settle`) slower than its own repeats because the clocks are still
# on their way up after 30 ms?  The same line with 30 / 100 / 300 / 1000 ms of settling.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step20
mkdir -p $OUT
cd $ROOT
for pass in 1 2 3; do
  for ms in 30 100 300 1000; do
    timeout 200 python bench.py --steps 20 --warmup 5 --settle-ms $ms --no-extras --no-cpu-baseline --no-pmc > $OUT/b.json 2>/dev/null
    python - <<PY $OUT/b.json $ms | tee -a $OUT/settle_sweep.log
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["repeats"]
print("settle_ms %s: cold %.2f headline %.2f repeats %.2f / %.2f / %.2f steady %.2f us" % (sys.argv[2], d["cold_start"]["ms_per_step"]*1e3, d["ms_per_step"]*1e3, r["ms_per_step_min"]*1e3, r["ms_per_step_median"]*1e3, r["ms_per_step_max"]*1e3, d["steady_state"]["ms_per_step_median"]*1e3))
PY
  done
done
