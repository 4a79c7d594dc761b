#!/bin/bash
# Round 6, GPU call: the bench line of BASELINE config 5's whole cluster (262 144 nodes x 16 384 pods) again on the shipped tree - the first
# one (tools/r06_c5whole.sh) was taken before the non-temporal verdict stores - with its kernel stats and the mode-B leg.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step24
mkdir -p $OUT
cd $ROOT
timeout 800 python bench.py --gpus 1 --config 5 --total-nodes 262144 --pods 16384 --steps 100 --warmup 20 --no-cpu-baseline > $OUT/bench_c5whole.json 2> $OUT/bench_c5whole.err
echo "rc=$?"
python - <<'PY' $OUT/bench_c5whole.json
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.4e ms_per_step %.4f steady %.4f" % (d["value"], d["ms_per_step"], d["steady_state"]["ms_per_step_median"]))
print({k:v for k,v in d["roofline"].items() if not isinstance(v,(str,dict))})
print("mode_b", {k:v for k,v in d.get("mode_b",{}).items() if k!="parity"}, (d.get("mode_b",{}).get("parity") or {}).get("identical"))
print("e2e", d.get("end_to_end")); print("single", d.get("single_find"))
PY
ST=$OUT/stats; rm -rf $ST; mkdir -p $ST
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ST -o s -- python $ROOT/bench.py --gpus 1 --config 5 --total-nodes 262144 --pods 16384 --steps 100 --warmup 20 --no-cpu-baseline --no-pmc --no-extras > $ST/run.log 2>&1)
find $ST -name "*kernel_stats.csv" -exec cp {} $OUT/c5whole_kernel_stats.csv \;
head -4 $OUT/c5whole_kernel_stats.csv | cut -c1-200
rm -rf $ST
