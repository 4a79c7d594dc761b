#!/bin/bash
# full GPU check of the round: parity suite, smoke, default bench, rocprofv3 kernel stats of the benchmarked command
mkdir -p gpurun_out/full
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/full/pytest.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/full/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/full/smoke.log
SECONDS=0; timeout 900 python bench.py > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err; echo "bench rc=$? seconds=$SECONDS" >> gpurun_out/full/bench.err
ROOT=$(pwd); OUT=$ROOT/gpurun_out/full/stats; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python $ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-pmc --no-extras > $OUT/run.log 2>&1)
find $OUT -name "*kernel_stats.csv" -exec cp {} gpurun_out/full/kernel_stats.csv \;
cat gpurun_out/full/pytest.log; tail -2 gpurun_out/full/smoke.log; tail -1 gpurun_out/full/bench.err; head -4 gpurun_out/full/kernel_stats.csv
