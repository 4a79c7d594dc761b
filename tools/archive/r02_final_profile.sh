#!/bin/bash
# Round-2 profile set for profiles/r02/: rocprofv3 kernel stats of the benchmarked command (fused step kernel), the
# per-role stand-alone times, and the PMC passes (one counter set per pass, never combined with tracing).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_final
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-pmc --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $BENCH > $OUT/stats.log 2>&1
NHDFIT_ROLE_KERNELS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/roles -o s -- $BENCH > $OUT/roles.log 2>&1
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- $BENCH > $OUT/pmc$i.log 2>&1
  NHDFIT_ROLE_KERNELS=1 timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/rpmc$i -o p -- $BENCH > $OUT/rpmc$i.log 2>&1
done
python $ROOT/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
python - <<PY >> $OUT/summary.txt
import collections, csv, glob
print("== per-role stand-alone kernels (NHDFIT_ROLE_KERNELS=1): role 0 choose, 1 shapes, 2 finish, 3 digest, 4 fit")
for f in glob.glob("$OUT/roles/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("  %-78s calls=%s avg_ns=%s" % (r["Name"][:78], r["Calls"], r["AverageNs"]))
for f in sorted(glob.glob("$OUT/rpmc*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        if "k_role" in row["Kernel_Name"]: acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in sorted(acc.items()):
        print("  ", k)
        for c, v in cs.items(): print("      %-24s mean=%.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
cp $OUT/stats/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/roles -name "*kernel_stats.csv" -exec cp {} $OUT/role_kernel_stats.csv \;
grep -A12 "k_step" $OUT/summary.txt | head -80
