#!/usr/bin/env python3
"""tools/recompute_roofline.py [TAG] -- every roofline fraction of the bench line recomputed from the files under profiles/ alone:
  launches, mean launch duration, union of the launch intervals   <- profiles/TAG_union.json   (rocprofv3 --kernel-trace, timed region)
  mean launch duration over the whole process                      <- profiles/TAG_kernel_stats.csv (rocprofv3 --stats)
  HBM traffic per launch, L2 hit rate                              <- profiles/TAG_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)
  algorithmic bytes per launch (unit counts x byte model)          <- profiles/TAG_bench_traced.json (the bench line of the traced run:
                                                                      ray / vertex / record counters of the device)
and compared with the fractions the bench line itself carries (from the library's hipEvents).  No GPU needed."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
P = lambda name: os.path.join(ROOT, "profiles", "%s_%s" % (tag, name))
PEAK = 8000.0  # GB/s, MI355X HBM3E
bench = json.loads([l for l in open(P("bench_traced.json")).read().splitlines() if l.startswith("{")][-1])
union = json.load(open(P("union.json")))["legs"][0]["kernels"]
traffic = json.load(open(P("traffic.json")))["kernels"]
whole = {}
for r in csv.DictReader(open(P("kernel_stats.csv"))):
    n = r["Name"]
    k = "k_shade" if n.startswith("k_shade") else "k_trace<closest>" if "k_trace<false, false" in n else "k_trace<any>" if "k_trace<true, false" in n else None
    if k:
        whole[k] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
print("%-18s %8s | %28s | %10s %10s | %21s | %21s | %15s" % ("kernel", "launches", "mean launch ms ev/trace/proc", "alg GB/l", "walked GB", "frac alg: bench / trace", "frac alg over union", "traffic GB, frac"))
worst = 0.0
for k, e in bench["roofline"]["kernels"].items():
    u, t = union[k], traffic.get(k, {})
    alg = e["algorithmic_bytes_per_launch"]
    frac_trace = alg / (u["avg_launch_ms"] * 1e-3) / 1e9 / PEAK
    frac_union = alg * u["launches"] / (u["union_ms"] * 1e-3) / 1e9 / PEAK
    tr = t.get("traffic_bytes_per_launch")
    frac_tr = tr / (u["avg_launch_ms"] * 1e-3) / 1e9 / PEAK if tr else float("nan")
    worst = max(worst, abs(e["avg_launch_ms"] / u["avg_launch_ms"] - 1))
    assert e["launches"] == u["launches"], (k, e["launches"], u["launches"])
    print("%-18s %8d | %8.3f / %7.3f / %7.3f | %10.2f %10.2f | %9.3f / %9.3f | %9.3f / %9.3f | %7.2f, %6.3f" % (
        k, u["launches"], e["avg_launch_ms"], u["avg_launch_ms"], whole[k][1], alg / 1e9, e.get("walked_tree_bytes_per_launch", 0) / 1e9,
        e["frac_per_launch"], frac_trace, e["frac_over_union"], frac_union, (tr or 0) / 1e9, frac_tr))
top = bench["roofline"]
print("top-level roofline of the bench line: kernel %s, achieved %.0f GB/s, frac %.3f (peak %.0f GB/s)" % (top["kernel"], top["achieved"], top["frac"], top["peak"]))
print("events vs trace, mean launch duration: largest difference %.1f %%" % (100 * worst))
