#!/bin/bash
# choose role lane-parallel (wavefront = tile, lane = shape; 8 blocks instead of 128) against a wavefront per shape
for v in 1 0 1 0; do
  echo "== NHDFIT_CHOOSE_LANES=$v"
  if [ $v = 1 ]; then export NHDFIT_CHOOSE_LANES=1; else unset NHDFIT_CHOOSE_LANES; fi
  timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-extras 2>&1 | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value']/1e12, o['ms_per_step'], o['roofline']['kernel_ms'], o['placed_pods'])"
done
export NHDFIT_CHOOSE_LANES=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "baseline_configs or golden or random_clusters or pipelined" 2>&1 | grep -E "passed|failed"
