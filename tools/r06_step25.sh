#!/bin/bash
# Round 6, GPU call: is the headline region slow because it is the FIRST short burst behind the long settling burst?  The same line with
# 0 / 1 / 3 untimed short regions at the end of the settling phase (experiment flag --settle-regions).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step25
mkdir -p $OUT
cd $ROOT
for pass in 1 2 3 4; do
  for r in 0 1 3; do
    timeout 200 python bench.py --steps 20 --warmup 5 --settle-regions $r --no-extras --no-cpu-baseline --no-pmc > $OUT/b.json 2>/dev/null
    python - <<PY $OUT/b.json $r | tee -a $OUT/settle_regions.log
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["repeats"]
print("settle_regions %s: cold %.2f headline %.2f repeats %.2f / %.2f / %.2f steady %.2f us" % (sys.argv[2], d["cold_start"]["ms_per_step"]*1e3, d["ms_per_step"]*1e3, r["ms_per_step_min"]*1e3, r["ms_per_step_median"]*1e3, r["ms_per_step_max"]*1e3, d["steady_state"]["ms_per_step_median"]*1e3))
PY
  done
done
