#!/bin/bash
# tools/bench_trace.sh TAG [bench args] -- rocprofv3 kernel trace of a short bench.py run; prints the launches after the last
# k_primary (the end of the render: what the caller waits for) and per-kernel totals.  GPU box only.
set -u
TAG=${1:-bench}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --no-cpu-baseline "$@" > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
tail -1 $OUT/trace.log | cut -c1-400
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
last = max(i for i, r in enumerate(rows) if "k_primary" in r[2])
t0 = rows[last][0]
print("kernels after the last k_primary longer than 0.3 ms, and every tail kernel longer than 0.05 ms (start ms, duration ms, name):")
agg = {}
for s, e, n in rows[last:]:
    short = n.split("(")[0][:40]
    agg.setdefault(short, [0, 0.0]); agg[short][0] += 1; agg[short][1] += (e - s) / 1e6
    if (e - s) > 300000 or ("k_tail" in n and (e - s) > 50000):
        print("  %9.3f %9.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, short))
print("totals after the last k_primary:", {k: (v[0], round(v[1], 2)) for k, v in agg.items()})
print("span ms:", (max(r[1] for r in rows[last:]) - t0) / 1e6)
PY
