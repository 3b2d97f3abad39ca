#!/bin/bash
# tools/busy_fraction.sh [bench args] -- rocprofv3 kernel trace of a short bench.py run: what share of the render's wall time has at
# least one kernel running, and how many run at once on average.  GPU box only.
set -u
OUT=gpurun_out/prof_busy
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 280 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python3 bench.py --no-cpu-baseline "$@" > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:30]))
rows.sort()
prim = [r[0] for r in rows if "k_primary" in r[2]]
# steady state: from the 6th k_primary to the last one
t0, t1 = prim[5], prim[-1]
ev = []
for s, e, n in rows:
    s, e = max(s, t0), min(e, t1)
    if e > s: ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = 0; area = 0; depth = 0; last = t0
for t, d in ev:
    if depth > 0: busy += t - last
    area += depth * (t - last)
    depth += d; last = t
print("steady state %.1f ms: some kernel running %.1f %% of the time, %.2f kernels at once on average" % ((t1 - t0) / 1e6, 100.0 * busy / (t1 - t0), area / (t1 - t0)))
gaps = []
depth = 0; last = t0
for t, d in ev:
    if depth == 0 and t - last > 20000: gaps.append(t - last)
    depth += d; last = t
print("idle gaps > 20 us: %d, %.1f ms in all, longest %.3f ms" % (len(gaps), sum(gaps) / 1e6, max(gaps) / 1e6 if gaps else 0))
PY
find $OUT -name "*kernel_trace.csv" -delete
