#!/bin/bash
# mode B after the taken-bit rewrite of k_seq: parity subset + timings for 16 and 8 pods per round
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mode_b or commit or sched or batched" 2>&1 | tail -5
for pods in 16; do
  echo "== NHDFIT_SEQ_PODS=$pods"
  NHDFIT_SEQ_PODS=$pods NHDFIT_SEQ_PROF=1 timeout 300 python tools/time_mode_b.py 65536 4096 4 2>&1 | tail -3
  NHDFIT_SEQ_PODS=$pods timeout 300 python tools/time_mode_b.py 65536 4096 4 2>&1 | tail -1
  NHDFIT_SEQ_PODS=$pods timeout 300 python tools/time_mode_b.py 65536 4096 5 2>&1 | tail -1
done
