#!/usr/bin/env python3
"""tools/pmc_tables.py PROF_DIR -- the derived figures of tools/profile_round.sh's pmc_ta / pmc_a / pmc_b / pmc_write / pmc_sq passes
per kernel: texture-addresser busy share, L1->L2 and fabric read latency, L2 hit rate, VALU lane utilisation."""
import collections, csv, glob, os, sys

d = sys.argv[1]


def name(n):
    return ("k_trace<closest>" if "k_trace<false, false" in n else "k_trace<any>" if "k_trace<true, false" in n else "k_shade" if "k_shade" in n
            else "k_tail_team" if "k_tail_team" in n else "k_tail_wave" if "k_tail_wave" in n else "k_tail" if "k_tail(" in n else "k_primary" if "k_primary" in n else None)


tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for sub in ("pmc_ta", "pmc_a", "pmc_b", "pmc_write", "pmc_sq", "pmc_sq2"):
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = name(r["Kernel_Name"])
            if k:
                tot[k][sub + ":" + r["Counter_Name"]] += float(r["Counter_Value"])
                calls[k].add((sub, r["Dispatch_Id"]))
for k in sorted(tot):
    t = collections.defaultdict(float, tot[k])
    g = lambda c: max(t[c], 1.0)
    print("%s" % k)
    print("   raw: " + "  ".join("%s %.4g" % (c.split(":")[1], v) for c, v in sorted(t.items())))
    print("   L2 hit rate %.3f (TCC_HIT / (TCC_HIT + TCC_MISS))" % (t["pmc_write:TCC_HIT_sum"] / max(1.0, t["pmc_write:TCC_HIT_sum"] + t["pmc_write:TCC_MISS_sum"])))
    print("   TA busy / GUI active %.2f; TCP pending-stall cycles %.3g" % (t["pmc_ta:TA_BUSY_avr"] / g("pmc_ta:GRBM_GUI_ACTIVE"), t["pmc_ta:TCP_PENDING_STALL_CYCLES_sum"]))
    print("   L1->L2 read latency %.0f cycles per request; L1 latency %.0f cycles per access; fabric read latency %.0f cycles per request; UTCL1 miss rate %.4f" % (
        t["pmc_a:TCP_TCC_READ_REQ_LATENCY_sum"] / g("pmc_ta:TCP_TCC_READ_REQ_sum"), t["pmc_b:TCP_TCP_LATENCY_sum"] / g("pmc_b:TCP_TOTAL_CACHE_ACCESSES_sum"),
        t["pmc_b:TCC_EA0_RDREQ_LEVEL_sum"] / g("pmc_b:TCC_EA0_RDREQ_sum"), t["pmc_a:TCP_UTCL1_TRANSLATION_MISS_sum"] / g("pmc_a:TCP_UTCL1_REQUEST_sum")))
    print("   VALU lane utilisation %.2f (SQ_THREAD_CYCLES_VALU / 64 / SQ_ACTIVE_INST_VALU); VALU instructions per wave %.0f; wave cycles per wave %.0f" % (
        t["pmc_sq:SQ_THREAD_CYCLES_VALU"] / 64.0 / g("pmc_sq2:SQ_ACTIVE_INST_VALU"), t["pmc_sq:SQ_INSTS_VALU"] / g("pmc_sq:SQ_WAVES"), t["pmc_sq:SQ_WAVE_CYCLES"] / g("pmc_sq:SQ_WAVES")))
