#!/bin/bash
# Round 5, second GPU call: the two-stage commit of mode B's chain (seq2_kernel.h commit_summary_wave / commit_picks_wave) and the
# fit role's lane order + plane-split pair table (step_kernel.h k_xorder, step_fit.h).  Whole GPU suite first, then the bench in
# the driver's form, the chain's phase log (tuning build), and the lane order on / off with the in-run counters.
#   gpurun --timeout 1100 -- bash tools/r05_step2.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_step2
mkdir -p $OUT
cd $ROOT
SECONDS=0
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$? seconds=$SECONDS" | tee -a $OUT/pytest_gpu.log
tail -n 5 $OUT/pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err
echo "bench rc=$? seconds=$SECONDS"
TL=$ROOT/nhd_amd/libnhdfit_tuning.so
{
for shape in "4096 256 2" "65536 4096 4" "16384 1024 3"; do
  echo "== $shape ship"; timeout 200 python tools/time_mode_b.py $shape 2>&1 | tail -1
  echo "== $shape phases (tuning build)"; NHDFIT_LIBRARY=$TL NHDFIT_SEQ_PROF=1 timeout 200 python tools/time_mode_b.py $shape 2>&1 | tail -4
done
} > $OUT/mode_b_phases.log 2>&1
echo "mode b seconds=$SECONDS"
# lane order on / off (tuning build reads NHDFIT_LANE_ORDER), counters collected in the run
for lo in 1 0; do
  NHDFIT_LIBRARY=$TL NHDFIT_LANE_ORDER=$lo timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_lane_order_$lo.json 2> $OUT/bench_lane_order_$lo.err
  echo "lane order $lo rc=$? seconds=$SECONDS"
done
python - <<'PY'
import json, glob, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05_step2")
for f in sorted(glob.glob(out + "/bench_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d["roofline"]
    print(os.path.basename(f), "ms/step %.5f" % d["ms_per_step"], "steady %.5f" % (d["steady_state"] or {}).get("ms_per_step_median", 0),
          "lds", r.get("lds"), "valu", (r.get("issue") or {}).get("valu_wave_insts_per_launch"), "kernel_ms", r.get("kernel_ms"))
    if "mode_b" in d:
        print("   mode_b", d["mode_b"]["decisions_per_s"], d["mode_b"].get("parity", {}).get("identical"),
              [(o["config"], o["pods"], round(o["ms_per_step"] * 1e3, 2), round(o["mode_b_decisions_per_s"]), o["mode_b_parity"]["identical"]) for o in d["other_configs"]])
        print("   single_find", d["single_find"]["ms_per_call_median"], "end_to_end", d["end_to_end"]["ms_per_call"], "big", d["big_pod_find"]["ms_per_call_median"])
PY
