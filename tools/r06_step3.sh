#!/bin/bash
# Round 6, GPU call 4: where a k_step launch's time goes (tuning build: role windows + per-block phases of the fit role at one step of a
# running pipeline; phases of the drain launch), the scheduler loop through the drop-in class, and one PC-sampling attempt (last: if the
# profiler cannot do it on this box nothing else is lost).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step3
mkdir -p $OUT
cd $ROOT
export NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so
for step in 40 41 400 401; do
  NHDFIT_ROLE_TIMES=$step timeout 200 python bench.py --steps 500 --warmup 20 --no-settle --no-extras --no-cpu-baseline --no-pmc 2>&1 >/dev/null | grep "nhdfit" >> $OUT/role_windows.log
done
cat $OUT/role_windows.log
NHDFIT_DRAIN_PROF=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc 2>&1 >/dev/null | grep "drain of" | sort | uniq -c | sort -rn | head -12 > $OUT/drain_phases.log
cat $OUT/drain_phases.log
unset NHDFIT_LIBRARY
timeout 300 python -c "
import json, bench
print(json.dumps(bench.sched_loop(4, 0)))" > $OUT/sched_loop_bench_leg.json 2>$OUT/sched_loop_bench_leg.err; cat $OUT/sched_loop_bench_leg.json
timeout 300 python tools/time_sched_loop.py > $OUT/sched_loop_with_standin_bookkeeping.json 2>/dev/null; cat $OUT/sched_loop_with_standin_bookkeeping.json
timeout 200 python tools/time_findnode.py > $OUT/findnode_latency.json 2>/dev/null; cut -c1-600 $OUT/findnode_latency.json
# PC sampling (beta): stochastic first, host trap second; each under its own short timeout
PC=$OUT/pcs; rm -rf $PC; mkdir -p $PC
for method in stochastic host_trap; do
  unit=cycles; interval=1048576; [ $method = host_trap ] && unit=time && interval=100
  (cd /tmp && export TMPDIR=/tmp && ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $interval --kernel-trace --output-format csv -d $PC/$method -o pcs -- python $ROOT/bench.py --steps 2000 --warmup 100 --no-settle --no-extras --no-cpu-baseline --no-pmc > $PC/$method.log 2>&1; echo "pc sampling $method rc=$?")
  tail -3 $PC/$method.log | cut -c1-300
  find $PC/$method -type f | head; 
done
du -sh $PC; find $PC -name "*.csv" -size +20M -exec gzip {} \; ; du -sh $PC
