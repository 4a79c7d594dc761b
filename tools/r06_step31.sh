#!/bin/bash
# Round 6, GPU call: mode B's sequencer without its (unread) real-time-counter reads in the shipped build - A/B at the BASELINE shapes,
# then the mode-B parity tests on the new form.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step31
mkdir -p $OUT
cd $ROOT
for pass in 1 2 3; do
  for v in ship notimers; do
    for s in "65536 4096 4" "4096 256 2" "16384 1024 3" "32768 2048 5" "262144 4096 5"; do
      NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_cand_$v.so timeout 200 python tools/time_mode_b.py $s 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'c%d %d x %d: %.0f decisions/s (%.3f ms)' % (d['config'], d['nodes'], d['pods'], d['decisions_per_s'], d['mode_b_ms']))" | tee -a $OUT/mode_b_sequencer_timers_ab.log
    done
  done
done
NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_cand_notimers.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "mode_b or schedule or sequential or commits_reproduce or pending_list or scheduler_loop or decide or every_form" > $OUT/parity_mode_b.log 2>&1
echo "parity (mode B, no timers) rc=$? $(grep -E 'passed|failed' $OUT/parity_mode_b.log | tail -1)"
