#!/usr/bin/env python3
"""Per-kernel averages of the counters in a rocprofv3 --pmc run directory (counter_collection.csv)."""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    n = re.sub(r"void |zkmi::|rocprim::ROCPRIM_\d+_NS::|detail::", "", r["Kernel_Name"])
    g = "<G2>" if "Fp2" in n else ("<G1>" if "FpParams" in n else "")
    k = re.match(r"[A-Za-z0-9_]+", n).group(0) + g
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r["Dispatch_Id"])
names = sorted({c for k in acc for c in acc[k]})
print("kernel".ljust(28), "disp".rjust(5), " ".join(c.rjust(18) for c in names))
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", acc[k].get(names[0], 0))):
    d = len(cnt[k])
    print(k.ljust(28), str(d).rjust(5), " ".join(("%.4g" % (acc[k][c] / d)).rjust(18) for c in names))
