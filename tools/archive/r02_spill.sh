#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lds_rows or across_shards or full_size or baseline_configs" 2>&1 | tail -8
timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-extras 2>&1 | tail -1 | cut -c1-400
