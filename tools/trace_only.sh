#!/bin/bash
# tools/trace_only.sh TAG -- kernel trace of a short bench run (environment inherited); prints the durations of the last closest-hit walks
set -u
TAG=${1:-t}
OUT=gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
python3 - $OUT/trace_kernel_trace.csv <<'PY'
import csv, sys
rows = sorted((int(r["Start_Timestamp"]), r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in csv.DictReader(open(sys.argv[1])))
for tag, name in (("closest", "k_trace<false, false>"), ("any", "k_trace<true, false>"), ("shade", "k_shade")):
    d = [x for _, n, x in rows if name in n]
    print(tag, "total %.1f ms;" % (sum(d) / 1e3), " ".join("%.0f" % x for x in d[-42:-14]))
PY
