#!/usr/bin/env python3
"""tools/trace_union.py PROF_DIR [OUT.json] -- from a rocprofv3 --kernel-trace of bench.py: per timed leg (bracketed by bench.py's
marker fill kernels: FillFunctor<short> before, FillFunctor<signed char> after) and per kernel the number of launches, the sum and
mean of the launch durations and the UNION of the launch intervals (two lanes' launches of one kernel overlap), plus how much of the
leg some kernel was running at all.  bench.py's roofline block carries the same figures from the library's own hipEvents: this is
the profiler's view of them."""
import csv, glob, json, os, re, sys


def short(n):
    n0 = n.split("(")[0]
    m = re.search(r"k_trace<(true|false), (true|false)", n)  # <ANY, COUNT, QBVH>
    if m:
        return "k_trace<%s>%s" % ("any" if m.group(1) == "true" else "closest", " (counting)" if m.group(2) == "true" else "")
    m = re.search(r"\b(k_\w+)", n0)
    return m.group(1) if m else n0[:48]


def union(iv):
    iv.sort()
    tot, hi = 0, -1
    for s, e in iv:
        if s > hi: tot += e - s; hi = e
        elif e > hi: tot += e - hi; hi = e
    return tot


rows = []
for f in glob.glob(os.path.join(sys.argv[1], "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [e for s, e, n in rows if "FillFunctor<short>" in n]
ends = [s for s, e, n in rows if "FillFunctor<signed char>" in n]
legs = list(zip(starts, ends))
out = {"legs": []}
names = ["headline (bench.py's timed region)", "mesh_view"]
for i, (t0, t1) in enumerate(legs):
    per = {}
    for s, e, n in rows:
        if s >= t0 and e <= t1 and "FillFunctor" not in n:
            per.setdefault(short(n), []).append((s, e))
    allk = [x for v in per.values() for x in v]
    leg = {"leg": names[i] if i < len(names) else "leg %d" % i, "wall_ms": (t1 - t0) / 1e6, "some_kernel_running_ms": union(list(allk)) / 1e6,
           "kernels_at_once": sum(e - s for s, e in allk) / max(1, union(list(allk))), "kernels": {}}
    for k, v in sorted(per.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
        tot = sum(e - s for s, e in v)
        leg["kernels"][k] = {"launches": len(v), "sum_of_launch_ms": tot / 1e6, "avg_launch_ms": tot / len(v) / 1e6, "union_ms": union(list(v)) / 1e6}
    out["legs"].append(leg)
    print("== %s: wall %.1f ms, some kernel running %.1f ms (%.1f %%), %.2f kernels at once on average" % (leg["leg"], leg["wall_ms"], leg["some_kernel_running_ms"],
          100 * leg["some_kernel_running_ms"] / max(1e-9, leg["wall_ms"]), leg["kernels_at_once"]))
    print("%-32s %8s %12s %12s %12s" % ("kernel", "launches", "sum ms", "avg ms", "union ms"))
    for k, v in leg["kernels"].items():
        print("%-32s %8d %12.2f %12.4f %12.2f" % (k, v["launches"], v["sum_of_launch_ms"], v["avg_launch_ms"], v["union_ms"]))
if not legs:
    print("no marker kernels found (bench.py brackets its timed regions with two fill kernels)")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
