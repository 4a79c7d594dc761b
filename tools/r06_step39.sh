#!/bin/bash
# Round 6, GPU call: what a mode-B call is made of - rocprofv3 kernel stats of tools/time_mode_b.py (config 4, config 2), ship and base builds.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step39
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in ship base; do
  lib=$ROOT/nhd_amd/libnhdfit_cand_$v.so; [ $v = ship ] && lib=$ROOT/nhd_amd/libnhdfit.so
  for s in "65536 4096 4" "4096 256 2"; do
    tag=${v}_$(echo $s | tr ' ' '_')
    NHDFIT_LIBRARY=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o s -- python $ROOT/tools/time_mode_b.py $s > $OUT/$tag.log 2>&1
    find $OUT/$tag -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $OUT/${tag}_kernel_stats.csv
    rm -rf $OUT/$tag
    echo "== $tag"; tail -1 $OUT/$tag.log | cut -c1-200; head -12 $OUT/${tag}_kernel_stats.csv | cut -c1-160
  done
done
