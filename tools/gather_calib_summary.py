#!/usr/bin/env python3
"""Reads a rocprofv3 --pmc FETCH_SIZE run of tools/gather_calib and prints, per kernel, FETCH_SIZE (KB -> bytes) against the bytes the kernel is known to
read: the factor by which the counter has to be multiplied for THAT access pattern.  Output: one JSON object (commit it under profiles/).
    gather_calib_summary.py <rocprof_dir> [known_bytes=4294967296]"""
import collections, csv, glob, json, re, sys

d = sys.argv[1]
known = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 32
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
tot, cnt = collections.defaultdict(float), collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "FETCH_SIZE":
        continue
    k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
    tot[k] += float(r["Counter_Value"])
    cnt[k].add(r["Dispatch_Id"])
out = {"known_bytes_per_launch": known, "kernels": {}}
for k in tot:
    b = tot[k] / len(cnt[k]) * 1024
    out["kernels"][k] = {"FETCH_SIZE_bytes_per_launch": int(b), "launches": len(cnt[k]), "counter_over_known": round(b / known, 4), "correction_factor": round(known / b, 4) if b else None}
print(json.dumps(out, indent=1))
