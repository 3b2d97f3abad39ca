#!/usr/bin/env python3
"""tools/per_bounce.py DIR -- per-bounce kernel durations of one render batch from a rocprofv3 kernel trace (tools/profile_gpu.sh)"""
import csv, sys, glob, os
d = sys.argv[1]
f = glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seq = []
for r in rows:
    n = r["Kernel_Name"]
    t0, t1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "k_primary" in n:
        seq.append({"t0": t0, "k": []})
    if seq and ("k_trace" in n or "k_shade" in n or "k_next" in n or "k_accumulate" in n):
        kind = "ext" if "k_trace<false" in n else "conn" if "k_trace<true" in n else "shade" if "k_shade" in n else "acc" if "k_acc" in n else "next"
        seq[-1]["k"].append((kind, (t1 - t0) / 1e3, t0, t1))
b = seq[int(sys.argv[2]) if len(sys.argv) > 2 else 2]
ks = b["k"]
ext = [x for x in ks if x[0] == "ext"]; conn = [x for x in ks if x[0] == "conn"]; sh = [x for x in ks if x[0] == "shade"]
wall = (ks[-1][3] - b["t0"]) / 1e3
busy = sum(x[1] for x in ks)
print("bounces %d  wall %.1f us  sum kernels %.1f us  gaps %.1f us" % (len(ext), wall, busy, wall - busy))
print("ext %.0f  shade %.0f  conn %.0f" % (sum(x[1] for x in ext), sum(x[1] for x in sh), sum(x[1] for x in conn)))
for i in list(range(0, 12)) + list(range(12, min(len(ext), 300), 16)):
    print("%3d ext %6.0f shade %6.0f conn %6.0f" % (i, ext[i][1], sh[i][1], conn[i][1]))
for lo, hi in ((0, 10), (10, 40), (40, 100), (100, 1000)):
    print("bounces %d-%d: ext %.0f shade %.0f conn %.0f" % (lo, hi, sum(x[1] for x in ext[lo:hi]), sum(x[1] for x in sh[lo:hi]), sum(x[1] for x in conn[lo:hi])))
