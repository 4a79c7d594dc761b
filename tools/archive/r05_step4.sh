#!/bin/bash
# Round 5, fourth GPU call: whole GPU suite (every failure listed), then the fit role with a second record in flight for the
# narrow tiles (-DNHDFIT_FIT_PREFETCH=2, libnhdfit_pf2.so) against the shipped build, alternating.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_step4
mkdir -p $OUT
cd $ROOT
SECONDS=0
timeout 1000 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$? seconds=$SECONDS" | tee -a $OUT/pytest_gpu.log
grep -E "big pods at scale|passed|failed|^FAILED|^ERROR" $OUT/pytest_gpu.log | tail -n 15
for rep in 1 2; do
  for lib in ship pf2; do
    L=; [ $lib = pf2 ] && L=$ROOT/nhd_amd/libnhdfit_pf2.so
    NHDFIT_LIBRARY=$L timeout 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc > $OUT/bench_${lib}_$rep.json 2> $OUT/bench_${lib}_$rep.err
    echo "$lib $rep rc=$? seconds=$SECONDS"
  done
done
python - <<'PY'
import json, glob, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05_step4")
for f in sorted(glob.glob(out + "/bench_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(os.path.basename(f), "ms/step %.5f" % d["ms_per_step"], "cold %.5f" % d["cold_start"]["ms_per_step"], "repeats", d["repeats"]["ms_per_step_median"], "steady %.5f" % d["steady_state"]["ms_per_step_median"])
PY
