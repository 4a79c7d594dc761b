#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/probe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
probe() { tag=$1; shift
  env NHDFIT_ROLE_KERNELS=1 "$@" timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o s -- python $ROOT/tools/fit_probe.py > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
  echo "$tag: $(grep 'k_role<512, 4>' $f | cut -d, -f2,4 | tr -d '"')  digest: $(grep 'k_role<512, 3>' $f | cut -d, -f4)"
}
probe base X=1
probe nobitmap PROBE_BITMAP=0
probe cpw2 NHDFIT_CPW=2,2,2,1
probe cpw4 NHDFIT_CPW=4,4,2,1
probe cpw16 NHDFIT_CPW=16,12,8,4
probe cpw32 NHDFIT_CPW=32,24,16,8
probe cfg2 PROBE_CFG=2
probe cfg5 PROBE_CFG=5 PROBE_N=32768
