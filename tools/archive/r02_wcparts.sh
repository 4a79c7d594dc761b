#!/bin/bash
# digest role: blocks per tile that share the CPU rows (4 = default; fewer blocks leave more block slots to the other roles)
for v in 2 4 2 4 1; do
  echo "== NHDFIT_WC_PARTS=$v"
  NHDFIT_WC_PARTS=$v timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-extras 2>&1 | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value']/1e12, o['ms_per_step'], o['roofline']['kernel_ms'], o['placed_pods'])"
done
NHDFIT_WC_PARTS=2 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "baseline_configs or golden or random_clusters or pipelined" 2>&1 | grep -E "passed|failed"
