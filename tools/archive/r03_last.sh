#!/bin/bash
# Round 3, last full check after the single-launch find: parity suite, smoke, default bench, rocprofv3 kernel stats of the
# benchmarked command (the separate PMC passes of tools/r03_full.sh are not repeated: k_step's code is byte-identical).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_last
mkdir -p $OUT
cd $ROOT
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_full.log 2>&1; echo "pytest rc=$? seconds=$SECONDS" | tee -a $OUT/pytest_full.log
grep -E "passed|failed|error" $OUT/pytest_full.log | tail -3
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
SECONDS=0; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? seconds=$SECONDS" | tee -a $OUT/bench.err
ST=$OUT/stats; rm -rf $ST; mkdir -p $ST
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ST -o s -- python $ROOT/bench.py --steps 400 --warmup 400 --no-cpu-baseline --no-pmc > $ST/run.log 2>&1)
find $ST -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -8 $OUT/kernel_stats.csv | cut -c1-200
python -c "
import json
o=json.load(open('$OUT/bench.json'))
print(o['value']/1e12, o['ms_per_step'], o['repeats'])
r=o['roofline']; print({k:r[k] for k in ('bound','frac','kernel_ms','concurrency','achieved_from_wall','traffic','unit_fracs')})
print(o['mode_b']); print(o['single_find']); print(o['end_to_end']); print(o['other_configs'])
"
