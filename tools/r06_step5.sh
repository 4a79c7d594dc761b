#!/bin/bash
# Round 6, GPU call 6: device timeline of the driver's 20-step region (rocprofv3 --kernel-trace: every k_step / k_map_tiles dispatch with
# start and end), fit blocks per tile and pipes on the tuning build, the drain's phases after the NIC-bit change.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step5
mkdir -p $OUT
cd $ROOT
TL=$OUT/tl; rm -rf $TL; mkdir -p $TL
(cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $TL -o t -- python $ROOT/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc > $TL/run.log 2>&1)
python - <<'PY' $TL $OUT/timeline_driver_form.txt
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
# regions = maximal runs of k_step dispatches closed by k_map_tiles; print the LAST FOUR regions of 20+ k_step launches
ks = [i for i, r in enumerate(rows) if "k_step" in r[2] or "k_map_tiles" in r[2]]
with open(sys.argv[2], "w") as o:
    o.write("# start_us end_us dur_us gap_to_prev_start_us queue kernel   (relative to the first dispatch shown; driver form: settle, then 5 + 20 steps, then 5 x 20 steps)\n")
    sel = [rows[i] for i in ks][-130:]
    t0 = sel[0][0]
    prev = None
    for s, e, n, q in sel:
        nm = "k_step" if "k_step" in n else "k_map_tiles"
        o.write("%9.2f %9.2f %7.2f %7.2f q%s %s\n" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, ((s - prev) / 1e3 if prev else 0.0), q, nm))
        prev = s
print(open(sys.argv[2]).read()[-4200:])
PY
rm -rf $TL
export NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so
show() { python - <<PY $1 "$2"
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "driver-form %.3f us  repeats med %.3f  steady %.3f  kernel_ms %.4f" % (d["ms_per_step"]*1e3, d["repeats"]["ms_per_step_median"]*1e3, d["steady_state"]["ms_per_step_median"]*1e3, d["roofline"]["kernel_ms"]))
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
}
for half in 0 2 3 6 7; do
  NHDFIT_FIT_HALF=$half timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc > $OUT/bench_half$half.json 2>/dev/null; show $OUT/bench_half$half.json "four fit blocks per tile for width classes (mask) $half:"
done
for pipes in 2 3; do
  NHDFIT_PIPES=$pipes NHDFIT_FIT_HALF=2 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc > $OUT/bench_half2_pipes$pipes.json 2>/dev/null; show $OUT/bench_half2_pipes$pipes.json "mask 2, pipes $pipes:"
done
NHDFIT_DRAIN_PROF=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc 2>&1 >/dev/null | grep "drain of" | head -4
