#!/bin/bash
# Round 3 full GPU check: parity suite, smoke, default bench (with counters and extras), rocprofv3 kernel stats + PMC passes of the
# benchmarked command -> gpurun_out/r03_full (copy what is to be judged into profiles/r03).   gpurun -- bash tools/r03_full.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_full
mkdir -p $OUT
cd $ROOT
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1 || { echo "smoke failed"; tail -5 $OUT/smoke.log; exit 1; }
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_full.log 2>&1; echo "pytest rc=$? seconds=$SECONDS" | tee -a $OUT/pytest_full.log
grep -E "passed|failed|error" $OUT/pytest_full.log | tail -3
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
SECONDS=0; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? seconds=$SECONDS" | tee -a $OUT/bench.err
ST=$OUT/stats; rm -rf $ST; mkdir -p $ST
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ST -o s -- python $ROOT/bench.py --steps 400 --warmup 400 --no-cpu-baseline --no-pmc --no-extras > $ST/run.log 2>&1)
find $ST -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $ST -name "*kernel_trace.csv" -exec sh -c 'head -400 {} > '$OUT'/kernel_trace_head.csv' \;
head -6 $OUT/kernel_stats.csv | cut -c1-220
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES" "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  D=$OUT/pmc_$name; rm -rf $D; mkdir -p $D
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $D -o p -- python $ROOT/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-pmc --no-extras > $D/run.log 2>&1)
done
python tools/summarize_profile.py $OUT > $OUT/pmc_summary.txt 2>&1 || true
tail -30 $OUT/pmc_summary.txt
python -c "
import json
o=json.load(open('$OUT/bench.json'))
print(o['value']/1e12, o['ms_per_step'], o['repeats'])
r=o['roofline']; print({k:r[k] for k in ('bound','frac','kernel_ms','concurrency','achieved_from_wall','traffic','unit_fracs')})
print(o['mode_b']); print(o['end_to_end']); print(o['other_configs']); print(o['cpu_baseline']['value'], o['cpu_baseline']['python_restatement'])
"
