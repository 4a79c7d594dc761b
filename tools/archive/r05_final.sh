#!/bin/bash
# Round 5 full GPU check: smoke, the whole parity suite, the bench line as the driver asks for it (--steps 20 --warmup 5, counters and
# extras on), rocprofv3 kernel stats (+ the longest k_step dispatches of the trace, by position) and one PMC pass per counter set of the
# benchmarked command -> gpurun_out/r05_final_<tag> (what is to be judged is copied into profiles/r05).
#   gpurun --timeout 1500 -- bash tools/r05_final.sh [tag]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-a}
OUT=$ROOT/gpurun_out/r05_final_$TAG
mkdir -p $OUT
cd $ROOT
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
SECONDS=0
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? seconds=$SECONDS" | tee -a $OUT/pytest_gpu.log
grep -E "big pods at scale|passed|failed|^FAILED|^ERROR" $OUT/pytest_gpu.log | tail -8
SECONDS=0; timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err; echo "bench rc=$? seconds=$SECONDS" | tee -a $OUT/bench_driver_form.err
python - <<'PY' $OUT/bench_driver_form.json
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "cold", d["cold_start"]["ms_per_step"], "steady", d["steady_state"]["ms_per_step_median"], "repeats", d["repeats"])
r=d["roofline"]; print("frac", r["frac"], "kernel_ms", r["kernel_ms"], "unit_fracs", r.get("unit_fracs"), "lds", r.get("lds"), "traffic", r.get("traffic"), "valu", (r.get("issue") or {}).get("valu_wave_insts_per_launch"))
print("mode_b", {k:v for k,v in d["mode_b"].items() if k!="parity"}, d["mode_b"]["parity"]["identical"])
print("single", d["single_find"]["ms_per_call_median"], "e2e", d["end_to_end"]["ms_per_call"], "score_only", d["score_only"]["ms_per_step"], "deltas", d["deltas"]["deltas_per_s"])
print("big", d["big_pod_find"])
for o in d["other_configs"]: print({k:(v if not isinstance(v,dict) else v.get("identical")) for k,v in o.items()})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("cores"), (d["cpu_baseline"].get("python_restatement") or {}).get("value"))
PY
ST=$OUT/stats; rm -rf $ST; mkdir -p $ST
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ST -o s -- python $ROOT/bench.py --steps 400 --warmup 400 --no-cpu-baseline --no-pmc --no-extras > $ST/run.log 2>&1)
find $ST -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -8 $OUT/kernel_stats.csv | cut -c1-200
# where in the run do the longest k_step dispatches sit?  (r04's kept trace had ONE of 20.4 ms among 12 402 of ~30 us)
python - <<'PY' $ST $OUT/longest_k_step_dispatches.txt
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t0 = rows[0][0] if rows else 0
ks = [(e - s, i, s - t0, n) for i, (s, e, n) in enumerate(rows) if "k_step" in n]
with open(sys.argv[2], "w") as o:
    o.write("# k_step dispatches of the kernel trace (bench.py --steps 400 --warmup 400 with settle and steady-state legs): %d; all dispatches: %d\n" % (len(ks), len(rows)))
    o.write("# duration_us, position among all dispatches of the process (0 = first), start_ms after the first dispatch, kernel\n")
    for d, i, s, n in sorted(ks, reverse=True)[:8]:
        o.write("%.1f %d %.3f %s\n" % (d / 1e3, i, s / 1e6, n[:60]))
    first = [x for x in ks if x[1] < 40]
    o.write("# the first k_step dispatches of the process: " + ", ".join("%.1f us (position %d)" % (d / 1e3, i) for d, i, s, n in sorted(first, key=lambda x: x[1])[:6]) + "\n")
print(open(sys.argv[2]).read())
PY
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES" "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  D=$OUT/pmc_$name; rm -rf $D; mkdir -p $D
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $D -o p -- python $ROOT/bench.py --steps 60 --warmup 20 --no-settle --no-cpu-baseline --no-pmc --no-extras > $D/run.log 2>&1)
done
python tools/summarize_profile.py $OUT > $OUT/pmc_summary.txt 2>&1 || true
grep -A12 "k_step" $OUT/pmc_summary.txt | head -60
# the per-counter directories hold every dispatch: keep the summaries only (gpurun_out is capped)
rm -rf $OUT/pmc_*/ $ST
timeout 200 python tools/time_single_find.py > $OUT/single_find_latency.json 2>/dev/null; tail -c 900 $OUT/single_find_latency.json
# the whole-batch call (nhdfit_find: one launch) on the BASELINE shapes, and the general path for requests under the kernel trace
timeout 200 python tools/time_batch_find.py "4:65536:4096,2:0:0,3:0:0,5:32768:2048,5:32768:16384" > $OUT/batch_find_latency.json 2>/dev/null; cat $OUT/batch_find_latency.json
BP=$OUT/bigprof; rm -rf $BP; mkdir -p $BP
(cd /tmp && export TMPDIR=/tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $BP -o big -- python $ROOT/tools/time_big_find.py > $OUT/big_find_latency.json 2> $BP/err.log)
find $BP -name "*kernel_stats.csv" -exec cp {} $OUT/big_kernel_stats.csv \;
head -5 $OUT/big_kernel_stats.csv | cut -c1-200; cut -c1-900 $OUT/big_find_latency.json
rm -rf $BP
