#!/bin/bash
# Round 6, GPU call 15: what one enqueue costs the host (tuning build, NHDFIT_ENQ_PROF) at config 2, 3 and 4.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step14
mkdir -p $OUT
cd $ROOT
export NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so
for cfg in "2 4096 256" "3 16384 1024" "4 65536 4096"; do
  set -- $cfg
  NHDFIT_ENQ_PROF=1 timeout 200 python bench.py --config $1 --total-nodes $2 --pods $3 --steps 200 --warmup 20 --no-extras --no-cpu-baseline --no-pmc > $OUT/bench_c$1.json 2> $OUT/enq_c$1.log
  echo "config $1: $(tail -3 $OUT/enq_c$1.log | tr '\n' ' ' | cut -c1-400)"
  python - <<PY $OUT/bench_c$1.json
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   ms_per_step %.4f steady %.4f" % (d["ms_per_step"], d["steady_state"]["ms_per_step_median"]))
PY
done
