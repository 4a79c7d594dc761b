#!/bin/bash
# blocks per tile of the XCD-aware fit items: default (8 / 16 by row width), 8 for every tile, 16 for every tile
for v in 0 1 2 0 1; do
  echo "== NHDFIT_XCD_K=$v"
  NHDFIT_XCD_K=$v timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-extras 2>&1 | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value']/1e12, o['ms_per_step'], o['roofline']['kernel_ms'], o['placed_pods'])"
done
