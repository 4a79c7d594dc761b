#!/bin/bash
# Round 6, GPU call: mode B - (lw) the fetcher starts a window no lower than the first chunk that still holds a node with GPUs nobody took; (run) the sequencer decides a run of parked pods with GPUs in registers.  A/B of base / lw / run / both (ship) at the BASELINE shapes,
# the sequencer's phases in the tuning build, then the mode-B parity tests and a soak on the new form.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step37
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "mode_b or schedule or sequential or commits_reproduce or pending_list or scheduler_loop or decide or every_form or edge_of" > $OUT/parity_mode_b.log 2>&1
echo "parity (mode B, ship) rc=$? $(grep -E 'passed|failed' $OUT/parity_mode_b.log | tail -1)"
NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_cand_park.so timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "mode_b or schedule or sequential or commits_reproduce or pending_list or scheduler_loop or decide or every_form or edge_of" > $OUT/parity_mode_b_park.log 2>&1
echo "parity (mode B, park) rc=$? $(grep -E 'passed|failed' $OUT/parity_mode_b_park.log | tail -1)"
for pass in 1 2; do
  for v in base lw run ship park; do
    lib=$ROOT/nhd_amd/libnhdfit_cand_$v.so; [ $v = ship ] && lib=$ROOT/nhd_amd/libnhdfit.so
    for s in "65536 4096 4" "4096 256 2" "16384 1024 3" "32768 2048 5" "32768 16384 5" "262144 4096 5"; do
      NHDFIT_LIBRARY=$lib timeout 300 python tools/time_mode_b.py $s 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'c%d %d x %d: %.0f decisions/s (%.3f ms)' % (d['config'], d['nodes'], d['pods'], d['decisions_per_s'], d['mode_b_ms']))" | tee -a $OUT/mode_b_low_water_and_runs_ab.log
    done
  done
done
NHDFIT_SEQ_PROF=1 NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so timeout 300 python tools/time_mode_b.py 65536 4096 4 2>&1 | grep -E "k_decide|decisions_per_s" | tail -6 > $OUT/mode_b_phases_low_water_and_runs.log
timeout 900 python tools/soak_mode_b_gpu.py 150 1000 > $OUT/soak_mode_b_gpu.log 2>&1
echo "soak (ship) rc=$? $(tail -1 $OUT/soak_mode_b_gpu.log)"
NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_cand_park.so timeout 900 python tools/soak_mode_b_gpu.py 150 1000 > $OUT/soak_mode_b_gpu_park.log 2>&1
echo "soak (park) rc=$? $(tail -1 $OUT/soak_mode_b_gpu_park.log)"
