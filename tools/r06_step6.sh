#!/bin/bash
# Round 6, GPU call 7: blocks per tile for the widest tiles (their six-fetch sweep is the launch's critical path), the device timeline of the
# driver's 20-step regions, mode B with the lean decision engine.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step6
mkdir -p $OUT
cd $ROOT
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
for pass in 1 2; do
for k8 in 1 2 3 4; do
  NHDFIT_XCD_K8=$k8 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc > $OUT/bench_k8_${k8}_$pass.json 2>/dev/null; show $OUT/bench_k8_${k8}_$pass.json "8 x $k8 fit blocks per W = 8 tile, pass $pass:"
done
done
for k8 in 2 4; do
  NHDFIT_PIPES=3 NHDFIT_XCD_K8=$k8 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc > $OUT/bench_k8_${k8}_p3.json 2>/dev/null; show $OUT/bench_k8_${k8}_p3.json "8 x $k8 per W = 8 tile, 3 pipes:"
done
NHDFIT_XCD_K8=2 NHDFIT_ROLE_TIMES=400 timeout 200 python bench.py --steps 500 --warmup 20 --no-settle --no-extras --no-cpu-baseline --no-pmc 2>&1 >/dev/null | grep "fit blocks\|role fit\|role digest" | head -6
unset NHDFIT_LIBRARY
TL=$OUT/tl; rm -rf $TL; mkdir -p $TL
(cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $TL -o t -- python $ROOT/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc > $TL/run.log 2>&1)
python - <<'PY' $TL $OUT/timeline_driver_form.txt
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
sel = [r for r in rows if "k_step" in r[2] or "k_map_tiles" in r[2]]
# regions: runs of dispatches closed by k_map_tiles; keep those with 15..30 k_step launches (the 5 + 20 region and the 20-step repeats)
regions, cur = [], []
for r in sel:
    cur.append(r)
    if "k_map_tiles" in r[2]:
        nxt_is_drain = False
        regions.append(cur); cur = []
merged = []
for reg in regions:            # two drains close a region (one per pipe): glue a region that is only a drain to its predecessor
    if merged and all("k_map_tiles" in x[2] for x in reg): merged[-1] += reg
    else: merged.append(reg)
with open(sys.argv[2], "w") as o:
    o.write("# driver-form regions (k_step launches closed by the drain launches); times in us relative to the region's first dispatch\n")
    for reg in merged:
        n = sum("k_step" in x[2] for x in reg)
        if not 15 <= n <= 30: continue
        t0 = reg[0][0]
        end_step = max(x[1] for x in reg if "k_step" in x[2]); end_all = max(x[1] for x in reg)
        o.write("region: %d k_step launches; first start -> last k_step end %.2f us (%.2f per step), -> last drain end %.2f us (%.2f per step)\n" % (n, (end_step - t0) / 1e3, (end_step - t0) / 1e3 / n, (end_all - t0) / 1e3, (end_all - t0) / 1e3 / n))
        prev = None
        for s, e, nm, q in reg:
            o.write("   %8.2f %8.2f dur %6.2f gap %6.2f q%s %s\n" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, ((s - prev) / 1e3 if prev else 0.0), q, "k_step" if "k_step" in nm else "k_map_tiles"))
            prev = s
txt = open(sys.argv[2]).read()
print("\n".join(l for l in txt.splitlines() if l.startswith("region")))
print(txt[-1900:])
PY
rm -rf $TL
for n in "65536 4096 4" "4096 256 2" "16384 1024 3" "32768 2048 5"; do timeout 300 python tools/time_mode_b.py $n 2>/dev/null | cut -c1-230; done
