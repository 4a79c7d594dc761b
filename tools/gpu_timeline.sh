#!/bin/bash
# Kernel timeline of a short pipelined run (run through gpurun): gpurun_out/tl_<tag>/kernel_trace.csv
# Usage: tools/gpu_timeline.sh <tag> [env assignments / bench args]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/tl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -o t -- python $ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline "$@" > $OUT/log.txt 2>&1
f=$(find $OUT/raw -name "*kernel_trace.csv" | head -1)
python - "$f" "$OUT/timeline.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as o:
    o.write("kernel,queue,start_us,end_us,dur_us\n")
    for r in rows:
        n = r["Kernel_Name"]
        for k in ("k_step", "k_fit_only", "k_build_asc", "k_map", "k_resolve", "k_nogpu"):
            if k in n: n = k; break
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        o.write(f"{n[:24]},{r.get('Queue_Id','')},{s/1e3:.1f},{e/1e3:.1f},{(e-s)/1e3:.1f}\n")
PY
rm -rf $OUT/raw
tail -1 $OUT/log.txt | cut -c1-300
