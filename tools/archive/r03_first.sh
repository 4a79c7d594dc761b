#!/bin/bash
# Round 3, first GPU call: parity suite (incl. mode B at c3 / c4 65536x4096 / c5 shard vs the independent oracle), the step
# launch with its argument block by pointer vs by value (tuning build), the full default bench line, kernel stats,
# mode-B phase times.        gpurun -- bash tools/r03_first.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_first
mkdir -p $OUT
cd $ROOT
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_full.log 2>&1; echo "pytest rc=$? seconds=$SECONDS" | tee -a $OUT/pytest_full.log
grep -E "passed|failed|error" $OUT/pytest_full.log | tail -3
B="python bench.py --no-cpu-baseline --no-pmc --no-extras"
line() { python -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value']/1e12, o['ms_per_step'], o['roofline']['kernel_ms'], o['placed_pods'], o.get('repeats'))"; }
TL=$ROOT/nhd_amd/libnhdfit_tuning.so
{
for rep in 1 2; do
  echo "== ship build (by pointer)"; timeout 300 $B 2>&1 | tail -1 | line
  echo "== tuning build, by pointer"; NHDFIT_LIBRARY=$TL timeout 300 $B 2>&1 | tail -1 | line
  echo "== tuning build, by value"; NHDFIT_LIBRARY=$TL NHDFIT_ARGS_BY_VALUE=1 timeout 300 $B 2>&1 | tail -1 | line
done
} 2>&1 | tee $OUT/args_ab.log
SECONDS=0; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? seconds=$SECONDS" | tee -a $OUT/bench.err
ST=$OUT/stats; rm -rf $ST; mkdir -p $ST
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ST -o s -- python $ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-pmc --no-extras > $ST/run.log 2>&1)
find $ST -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -6 $OUT/kernel_stats.csv
NHDFIT_LIBRARY=$TL NHDFIT_SEQ_PROF=1 timeout 300 python tools/time_mode_b.py 65536 4096 4 2>&1 | tail -5 | tee $OUT/mode_b_c4.log
NHDFIT_LIBRARY=$TL NHDFIT_ROLE_TIMES=30 timeout 300 $B 2>&1 | grep "nhdfit\]" | tee $OUT/roles.log
tail -c 1500 $OUT/bench.json
