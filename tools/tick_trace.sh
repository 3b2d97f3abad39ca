#!/bin/bash
# tools/tick_trace.sh [W H] -- rocprofv3 kernel trace of tools/tick_rate.py; prints every kernel of the last path-traced tick. GPU box only.
set -u
OUT=gpurun_out/prof_tick
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python3 tools/tick_rate.py ${1:-1280} ${2:-720} 12 > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
tail -2 $OUT/trace.log
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:36]))
rows.sort()
prim = [i for i, r in enumerate(rows) if "k_primary" in r[2]]
a, b = prim[-2], prim[-1]
t0 = rows[a][0]
print("one tick = %.3f ms between two k_primary starts" % ((rows[b][0] - t0) / 1e6))
for s, e, n in rows[a:b]:
    print("  %8.3f %7.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, n))
PY
find $OUT -name "*kernel_trace.csv" -delete
