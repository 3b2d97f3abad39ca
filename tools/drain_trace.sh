#!/bin/bash
# tools/drain_trace.sh TAG [tail_team] -- rocprofv3 kernel trace of one render + drain (tools/drain_one.py): the sequence of
# k_tail / k_tail_team launches with their durations shows where the end of a render goes, round by round.  GPU box only.
set -u
TAG=${1:-drain}; TEAM=${2:-1}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python3 tools/drain_one.py $TEAM > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last render's drain: everything after the last k_primary
last = max(i for i, r in enumerate(rows) if "k_primary" in r[2])
t0 = rows[last][0]
print("kernels after the last k_primary (start ms, duration ms, name):")
agg = {}
for s, e, n in rows[last:]:
    short = n.split("(")[0][:40]
    agg.setdefault(short, [0, 0.0]); agg[short][0] += 1; agg[short][1] += (e - s) / 1e6
    if (e - s) > 200000 or "k_tail" in n:
        print("  %9.3f %9.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, short))
print("totals:", {k: (v[0], round(v[1], 2)) for k, v in agg.items()})
print("span ms:", (max(r[1] for r in rows[last:]) - t0) / 1e6)
PY
