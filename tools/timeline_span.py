#!/usr/bin/env python3
"""Kernel timeline of the LAST `span_ms` of a rocprofv3 kernel_trace.csv: start offset, duration, stream, kernel; plus the busy / idle summary.
usage: timeline_span.py <dir> <span_ms> [min_us=50]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
span = float(sys.argv[2]) * 1e6
min_ns = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 50e3
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tend = max(int(r["End_Timestamp"]) for r in rows)
t0 = tend - span
def short(n):
    n = re.sub(r"void |zkmi::|rocprim::ROCPRIM_\d+_NS::|detail::", "", n)
    g = "<G2>" if "Fp2" in n else ("<G1>" if "FpParams" in n else "")
    if "onesweep" in n: return "radix_sort"
    if "trampoline" in n: return "rocprim"
    return re.match(r"[A-Za-z0-9_]+", n).group(0) + g
iv = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e < t0: continue
    iv.append((s, e))
    if e - s >= min_ns:
        print("%8.3f ms  +%7.3f ms  s%-3s %s" % ((s - t0) / 1e6, (e - s) / 1e6, r["Stream_Id"], short(r["Kernel_Name"])))
iv.sort()
busy, cur_s, cur_e = 0, None, None
gaps = []
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, cur_e - t0))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("span %.2f ms, some kernel running %.2f ms, idle %.2f ms; largest gaps (ms @ offset): %s" % (span / 1e6, busy / 1e6, (span - busy) / 1e6,
      ", ".join("%.2f@%.1f" % (g / 1e6, o / 1e6) for g, o in sorted(gaps, reverse=True)[:10])))
