#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_final
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "baseline_configs or pipelined or full_size or golden" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -3
SECONDS=0; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? seconds=$SECONDS" | tee -a $OUT/bench.err
python -c "
import json
o=json.load(open('$OUT/bench.json'))
print(o['value']/1e12, o['ms_per_step'], o['repeats'])
for r in o['other_configs']: print(r)
print(o['end_to_end']); print(o['mode_b']['decisions_per_s'], o['mode_b']['parity']['identical'])
"
TL=$ROOT/nhd_amd/libnhdfit_tuning.so
NHDFIT_LIBRARY=$TL NHDFIT_ROLE_TIMES=500 timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-extras 2>&1 | grep "nhdfit\]" | tee $OUT/roles.log
timeout 300 python tools/time_findnode.py 2>/dev/null > $OUT/findnode_latency.json; cut -c1-400 $OUT/findnode_latency.json
timeout 300 python tools/time_release.py 2>/dev/null > $OUT/release_latency.json; cut -c1-300 $OUT/release_latency.json
