#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/probe
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or baseline or pipelined or edge or random" > $OUT/tests.log 2>&1; grep -E "passed|failed" $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
probe() { tag=$1; shift
  env NHDFIT_ROLE_KERNELS=1 "$@" timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o s -- python $ROOT/tools/fit_probe.py > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
  python - <<PY
import csv
rows={r["Name"]:r for r in csv.DictReader(open("$f"))}
out=[]
for k,v in rows.items():
    if "k_role" in k: out.append("%s=%.1f" % (k.split("k_role<512, ")[1][0], float(v["AverageNs"])/1e3))
print("$tag", " ".join(sorted(out)), "(role us: 0 choose 1 shapes 2 finish 3 digest 4 fit)")
PY
}
fused() { tag=$1; shift
  env "$@" timeout 90 python $ROOT/bench.py --steps 400 --warmup 20 --no-cpu-baseline > $OUT/f_$tag.json 2> $OUT/f_$tag.err
  python -c "import json; j=json.load(open('$OUT/f_$tag.json')); print('fused $tag', round(j['value']/1e12,3), 'T evals/s', round(j['ms_per_step']*1e3,1), 'us/step kernel', round(j['roofline']['kernel_ms']*1e3,1))" || tail -3 $OUT/f_$tag.err
}
probe map PROBE_MAP=1
probe t768 NHDFIT_FIT_BLOCKS=768
probe t1024 NHDFIT_FIT_BLOCKS=1024
probe t2304 NHDFIT_FIT_BLOCKS=2304
fused default X=1
fused t768 NHDFIT_FIT_BLOCKS=768
fused t1024 NHDFIT_FIT_BLOCKS=1024
fused t2304 NHDFIT_FIT_BLOCKS=2304
fused b256 NHDFIT_BLOCK=256
