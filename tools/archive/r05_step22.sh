#!/bin/bash
# Round 5: mode B after the Python wrapper's per-pod loop went vectorised (nhd_amd/engine.py schedule_batch) - rates and a parity subset.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_step22
mkdir -p $OUT
cd $ROOT
for shape in "65536 4096 4" "16384 1024 3" "32768 2048 5" "32768 16384 5" "4096 256 2"; do
  timeout 120 python tools/time_mode_b.py $shape 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], d['nodes'], d['pods'], round(d['decisions_per_s']), round(d['mode_b_ms'],3))" | tee -a $OUT/mode_b.log
done
timeout 600 python -m pytest tests -m gpu -x -q -k "mode_b_at_baseline or device_commits or commit_without_closure or mode_b_sequential" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 $OUT/pytest.log | cut -c1-200
