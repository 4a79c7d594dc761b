#!/bin/bash
# Round 6, GPU call 3: winner-test skip A/B, the whole GPU suite on the new default library, pipes sweep on the tuning build, mode B on
# config 5's whole cluster (the decision engine with span-limited node bit maps).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step2
mkdir -p $OUT
cd $ROOT
bash tools/r06_fit_ab.sh "noskip skip" skip | grep -E "parity|pass"
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? seconds=$SECONDS" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|^FAILED|^ERROR|^E  " $OUT/pytest_gpu.log | tail -8
for n in "262144 4096 5" "262144 16384 5" "32768 16384 5" "65536 4096 4"; do timeout 300 python tools/time_mode_b.py $n 2>/dev/null | tee -a $OUT/mode_b.log | cut -c1-260; done
export NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so
for pipes in 2 3; do
  NHDFIT_PIPES=$pipes timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc > $OUT/bench_pipes$pipes.json 2>/dev/null
  python - <<PY $OUT/bench_pipes$pipes.json $pipes
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("pipes", sys.argv[2], "driver-form %.3f us  repeats med %.3f  steady %.3f  kernel_ms %.4f" % (d["ms_per_step"]*1e3, d["repeats"]["ms_per_step_median"]*1e3, d["steady_state"]["ms_per_step_median"]*1e3, d["roofline"]["kernel_ms"]))
PY
done
