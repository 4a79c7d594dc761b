#!/bin/bash
# XCD-aware fit work items on / off: step time (no extras), twice each
for v in 1 0 1 0; do
  echo "== NHDFIT_XCD_ITEMS=$v"
  NHDFIT_XCD_ITEMS=$v timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-extras 2>&1 | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value']/1e12, o['ms_per_step'], o['roofline']['kernel_ms'], o['placed_pods'])"
done
