#!/bin/bash
# Round 5, GPU call: mode B with one clearing launch and one upload in front of the pass - parity subset, rates.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_step20
mkdir -p $OUT
cd $ROOT
SECONDS=0
for rep in 1 2; do
  for shape in "65536 4096 4" "16384 1024 3" "32768 2048 5" "32768 16384 5" "4096 256 2"; do
    timeout 120 python tools/time_mode_b.py $shape 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], d['nodes'], d['pods'], round(d['decisions_per_s']), round(d['mode_b_ms'],3))" | tee -a $OUT/mode_b.log
  done
done
echo "timing seconds=$SECONDS"
timeout 900 python -m pytest tests -m gpu -x -q -k "mode_b or schedule or seq or decide or pending or heterogeneous or commit or rank_to_rank or shards or scheduler_loop" > $OUT/pytest.log 2>&1
echo "pytest rc=$? seconds=$SECONDS"; tail -3 $OUT/pytest.log | cut -c1-300
