#!/usr/bin/env python3
"""tools/traffic.py PROF_DIR OUT.json -- HBM-side traffic per launch of the BVH-walk kernels from the rocprofv3 --pmc passes
of tools/profile_gpu.sh (FETCH_SIZE and WRITE_SIZE are collected in separate runs; both are in KiB).

Correction as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE reports half of the bytes of 16-byte-per-lane loads
(128-B requests tallied at 64 B), so it is doubled; WRITE_SIZE is taken as is.  Infinity-Cache hits are included in
both (the counters sit on the L2's fabric side), so this is an upper bound of DRAM traffic."""
import csv, glob, json, os, sys
from collections import defaultdict


dur = defaultdict(dict)  # kernel -> dispatch -> ms, as the counter passes saw them (counter collection serialises the kernels)


def sums(d, sub, counter):
    tot, calls = defaultdict(float), defaultdict(set)
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            n = r["Kernel_Name"]
            k = ("k_trace<closest>" if "k_trace<false, false" in n else "k_trace<any>" if "k_trace<true, false" in n else "k_shade" if "k_shade" in n else "k_primary" if "k_primary" in n
                 else "k_tail" if "k_tail" in n else "k_accumulate" if "k_accumulate" in n else None)
            if k:
                tot[k] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
                if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                    dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    return {k: (tot[k], len(calls[k])) for k in tot}


d, out = sys.argv[1], sys.argv[2]
command = sys.argv[3] if len(sys.argv) > 3 else "bench.py --steps 3 --warmup 1 --no-cpu-baseline"
fetch, write = sums(d, "pmc_fetch", "FETCH_SIZE"), sums(d, "pmc_write", "WRITE_SIZE")
hit, miss = sums(d, "pmc_write", "TCC_HIT_sum"), sums(d, "pmc_write", "TCC_MISS_sum")
res = {"source": os.path.basename(d.rstrip("/")), "command": command, "unit": "bytes per launch",
       "correction": "2 x FETCH_SIZE (gfx950, 16-B/lane loads) + WRITE_SIZE, KiB -> bytes", "kernels": {}}
for k in fetch:
    f, nf = fetch[k]; w, nw = write.get(k, (0.0, nf))
    res["kernels"][k] = {"launches": nf, "fetch_kib_raw": f, "write_kib_raw": w, "traffic_bytes_per_launch": (2 * f / nf + w / max(nw, 1)) * 1024, "command": command}
    if dur.get(k):
        res["kernels"][k]["avg_launch_ms"] = sum(dur[k].values()) / len(dur[k])  # in the counter passes: each kernel alone on the chip
    if k in hit and k in miss and hit[k][0] + miss[k][0] > 0:  # L2 (TCC) requests served by the L2 itself, over all of the kernel's launches
        res["kernels"][k]["l2_hit_rate"] = hit[k][0] / (hit[k][0] + miss[k][0])
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["kernels"], indent=1))
