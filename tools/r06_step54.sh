#!/bin/bash
# Round 6, GPU call: the whole-matrix tests with the oracle's teams sized by the usable cores (16 on this pool's boxes, not os.cpu_count()'s 256).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step54
mkdir -p $OUT
cd $ROOT
python -c "
import sys, os; sys.path.insert(0, '.')
from oracle import coracle
print('os.cpu_count', os.cpu_count(), 'usable', coracle.usable_cpus(), 'cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else None)" | tee $OUT/cpus.log
timeout 800 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x --timeout=400 --durations=6 -k "full_size or baseline_configs_vs_c_oracle or mode_b_at_baseline_sizes" > $OUT/pytest.log 2>&1; echo "rc=$? $(grep -E 'passed|failed' $OUT/pytest.log | tail -1)"
grep -E "s call" $OUT/pytest.log | head -8
