#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/probe2
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
probe() { tag=$1; shift
  env NHDFIT_ROLE_KERNELS=1 NHDFIT_FIT_BLOCKS=768 "$@" timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o s -- python $ROOT/tools/fit_probe.py > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
  python - <<PY
import csv
rows={r["Name"]:r for r in csv.DictReader(open("$f"))}
out=[]
for k,v in rows.items():
    if "k_role" in k: out.append("%s=%.1f" % (k.split("k_role<512, ")[1][0], float(v["AverageNs"])/1e3))
print("$tag", " ".join(sorted(out)))
PY
}
probe nothing NHDFIT_FIT_SKIP=15 PROBE_BITMAP=0
probe nostage NHDFIT_FIT_SKIP=31 PROBE_BITMAP=0
probe noepi NHDFIT_FIT_SKIP=47 PROBE_BITMAP=0
probe noloop NHDFIT_FIT_SKIP=79 PROBE_BITMAP=0
probe empty NHDFIT_FIT_SKIP=127 PROBE_BITMAP=0
probe empty1536 NHDFIT_FIT_SKIP=127 PROBE_BITMAP=0 NHDFIT_FIT_BLOCKS=1536
