#!/bin/bash
# tools/pmc_ta.sh TAG -- one rocprofv3 --pmc pass over bench.py with texture-addresser / L1 (TA, TCP) counters: is the BVH walk
# bound by the per-CU L1 path (64 divergent lanes x 4 dwordx4 loads per node step)?
set -u
TAG=${1:-ta}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null

BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -s KILL 300 rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_ta -o pmc -- $BENCH > $OUT/pmc_ta.log 2>&1 || { echo "pmc ta failed"; tail -5 $OUT/pmc_ta.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, os, collections
d = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for f in glob.glob(os.path.join(d, "pmc_ta", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "k_trace<closest>" if "k_trace<false, false>" in n else "k_trace<any>" if "k_trace<true, false>" in n else "k_shade" if "k_shade" in n else "k_tail" if "k_tail" in n else None
        if k: tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
for k in tot:
    t = tot[k]; g = max(t.get("GRBM_GUI_ACTIVE", 1), 1)
    print("%-18s launches %4d  " % (k, len(calls[k])) + "  ".join("%s %.4g" % (c, v) for c, v in sorted(t.items())) + "  | TA_BUSY_avr/GUI_ACTIVE %.2f" % (t.get("TA_BUSY_avr", 0) / g))
PY
