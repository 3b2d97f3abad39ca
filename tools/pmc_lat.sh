#!/bin/bash
# tools/pmc_lat.sh TAG -- two rocprofv3 --pmc passes over a short bench run: address translation (UTCL1) and load latencies
# seen by the L1 (TCP) and by the L2's fabric side (TCC_EA0 occupancy / requests), per kernel.  Environment is inherited.
set -u
TAG=${1:-lat}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -s KILL 300 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_a -o pmc -- $BENCH > $OUT/pmc_a.log 2>&1 || { echo "pass a failed"; tail -5 $OUT/pmc_a.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmc_b -o pmc -- $BENCH > $OUT/pmc_b.log 2>&1 || { echo "pass b failed"; tail -5 $OUT/pmc_b.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, os, collections
d = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for sub in ("pmc_a", "pmc_b"):
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            k = "k_trace<closest>" if "k_trace<false, false>" in n else "k_trace<any>" if "k_trace<true, false>" in n else "k_shade" if "k_shade" in n else None
            if k: tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
for k in tot:
    t = tot[k]
    print("%-18s " % k + "  ".join("%s %.4g" % (c, v) for c, v in sorted(t.items())))
    print("   UTCL1 miss rate %.4f; L1->L2 read latency %.0f cycles per request; L1 latency %.0f cycles per access; fabric read latency %.0f cycles per request" % (
        t["TCP_UTCL1_TRANSLATION_MISS_sum"] / max(t["TCP_UTCL1_REQUEST_sum"], 1), t["TCP_TCC_READ_REQ_LATENCY_sum"] / max(t["TCP_TCC_READ_REQ_sum"], 1),
        t["TCP_TCP_LATENCY_sum"] / max(t["TCP_TOTAL_CACHE_ACCESSES_sum"], 1), t["TCC_EA0_RDREQ_LEVEL_sum"] / max(t["TCC_EA0_RDREQ_sum"], 1)))
PY
