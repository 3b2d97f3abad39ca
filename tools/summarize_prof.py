#!/usr/bin/env python3
"""Summarise rocprofv3 csv output of tools/profile_gpu.sh: per-kernel time (kernel trace), per-kernel PMC sums, and the
counters of the largest dispatches of the BVH-walk kernels (the first bounces; the tail bounces are tiny)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

KERNELS = ("k_primary", "k_trace_rays", "k_trace", "k_tail_team", "k_tail_wave", "k_tail_decide", "k_tail", "k_carry_in", "k_carry_out", "k_fold_late", "k_shade", "k_accumulate", "k_finalize", "k_whitted", "k_next_bounce", "k_copy_hits",
           "k_gamma", "k_chroma", "k_vignette")


def short(name):
    n = name.split("(")[0]
    for k in KERNELS:
        if re.search(r"\b%s\b" % k, n):
            tmpl = ""
            if "<" in name:
                tmpl = name[name.index("<"):name.index(">") + 1]
            return k + tmpl
    return n[:60]


def main(out, command="bench.py --steps 3 --warmup 1"):
    files = glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True)
    agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    per = defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            k = short(r["Kernel_Name"])
            a = agg[k]
            a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
            per[k].append((d, r.get("Grid_Size", "?"), r.get("VGPR_Count", "?"), r.get("LDS_Block_Size", "?")))
    total = sum(a[1] for a in agg.values()) or 1
    print("== kernel trace (rocprofv3 --kernel-trace --stats) of `%s` (incl. counting pass + warmup) ==" % command)
    print("%-28s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-28s %8d %12.1f %10.2f %10.2f %10.1f %6.1f" % (k, a[0], a[1], a[1] / a[0], a[2], a[3], 100 * a[1] / total))
    print("\n== longest dispatches (us, grid, vgpr, lds) ==")
    for k in sorted(per):
        if k.startswith("k_trace") or k.startswith("k_shade") or k.startswith("k_primary") or k.startswith("k_tail"):
            top = sorted(per[k], reverse=True)[:4]
            print("%-28s " % k + "  ".join("%.0f/%s/%s/%s" % t for t in top))
    for sub in sorted(os.listdir(out)):
        if not sub.startswith("pmc_") or not os.path.isdir(os.path.join(out, sub)):
            continue
        files = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)
        cnt = defaultdict(lambda: defaultdict(float))
        disp = defaultdict(lambda: defaultdict(float))
        dk = {}
        calls = defaultdict(set)
        for f in files:
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
                calls[k].add(r["Dispatch_Id"])
                disp[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
                dk[r["Dispatch_Id"]] = k
        if not cnt:
            continue
        names = sorted({c for v in cnt.values() for c in v})
        print("\n== %s: counter sums over all dispatches ==" % sub)
        print("%-28s %8s " % ("kernel", "calls") + " ".join("%18s" % n for n in names))
        for k, v in sorted(cnt.items(), key=lambda kv: -sum(kv[1].values())):
            print("%-28s %8d " % (k, len(calls[k])) + " ".join("%18.0f" % v.get(n, 0) for n in names))
        print("-- %s: the two largest dispatches of each BVH-walk / shade kernel --" % sub)
        for k in sorted(cnt):
            if not (k.startswith("k_trace<") or k.startswith("k_shade")):
                continue
            ds = sorted((d for d in disp if dk[d] == k), key=lambda d: -sum(disp[d].values()))[:2]
            for d in ds:
                print("%-28s %8s " % (k, "#" + d) + " ".join("%18.0f" % disp[d].get(n, 0) for n in names))


if __name__ == "__main__":
    main(*sys.argv[1:3])
