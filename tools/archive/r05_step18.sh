#!/bin/bash
# Round 5, GPU call 18: mode B - list entries per fetcher (16 shipped; 32 / 64 = the round's earlier form), parity subset on the shipped build.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_step18
mkdir -p $OUT
cd $ROOT
SECONDS=0
for lib in libnhdfit.so libnhdfit_chunk32.so libnhdfit_chunk64.so libnhdfit.so; do
  echo "== $lib" | tee -a $OUT/mode_b_fetch_chunk.log
  for shape in "65536 4096 4" "16384 1024 3" "32768 2048 5" "32768 16384 5" "4096 256 2"; do
    NHDFIT_LIBRARY=$ROOT/nhd_amd/$lib timeout 120 python tools/time_mode_b.py $shape 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], d['nodes'], d['pods'], round(d['decisions_per_s']), round(d['mode_b_ms'],3))" | tee -a $OUT/mode_b_fetch_chunk.log
  done
done
echo "timing seconds=$SECONDS"
timeout 700 python -m pytest tests -m gpu -x -q -k "mode_b or schedule or seq or decide or pending or heterogeneous" > $OUT/pytest.log 2>&1
echo "pytest rc=$? seconds=$SECONDS"; tail -3 $OUT/pytest.log | cut -c1-300
