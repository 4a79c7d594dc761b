#!/bin/bash
# Round 5, last GPU call: smoke + a parity subset on the tree after the clean-up, and the single-launch finds under the kernel trace.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_step21
mkdir -p $OUT
cd $ROOT
SECONDS=0
timeout 200 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -x -q -k "single_launch or mode_b_at_baseline or schedule or golden_vectors or pipelined or big_pods_at" > $OUT/pytest.log 2>&1
echo "pytest rc=$? seconds=$SECONDS"; tail -3 $OUT/pytest.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o f -- python $ROOT/tools/time_batch_find.py "4:65536:4096,2:0:0,3:0:0,5:32768:2048" > $OUT/batch_find_traced.json 2> $OUT/err.log
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/find_kernel_stats.csv && head -6 $OUT/find_kernel_stats.csv | cut -c1-220
python - <<'PY' $OUT/prof $OUT/findn_dispatches.txt
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_findn" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Grid_Size", 0) or 0)))
rows.sort()
by = {}
for s, e, g in rows: by.setdefault(g, []).append((e - s) / 1e3)
with open(sys.argv[2], "w") as o:
    o.write("# k_findn dispatches of tools/time_batch_find.py by grid size (threads): count, min / median / max duration in us\n")
    for g, v in sorted(by.items()):
        v.sort(); o.write("%d threads: %d dispatches, %.1f / %.1f / %.1f us\n" % (g, len(v), v[0], v[len(v) // 2], v[-1]))
print(open(sys.argv[2]).read())
PY
rm -rf $OUT/prof
