#!/bin/bash
# A/B of a kernel variant: default step time, three times, plus parity of the BASELINE shapes
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-extras 2>&1 | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value']/1e12, o['ms_per_step'], o['roofline']['kernel_ms'], o['placed_pods'])"
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "baseline_configs" 2>&1 | grep -E "passed|failed"
