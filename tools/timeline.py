#!/usr/bin/env python3
"""Prints the kernel timeline of the LAST proof in a rocprofv3 kernel_trace.csv (start offset, duration, stream)."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last proof starts at the last k_msm_digits-before-... find the last '__amd_rocclr_copyBuffer' gap; simpler: last 2 digits kernels
idx = [i for i, r in enumerate(rows) if "k_msm_digits" in r["Kernel_Name"]]
start_i = idx[-2]
# walk back to include the copy kernels just before
while start_i > 0 and int(rows[start_i]["Start_Timestamp"]) - int(rows[start_i - 1]["End_Timestamp"]) < 200000 and "k_reduce_wave" not in rows[start_i - 1]["Kernel_Name"]:
    start_i -= 1
t0 = int(rows[start_i]["Start_Timestamp"])
def short(n):
    n = re.sub(r"void |zkmi::|rocprim::ROCPRIM_\d+_NS::|detail::", "", n)
    g = "<G2>" if "Fp2" in n else ("<G1>" if "FpParams" in n else "")
    if "onesweep" in n: return "radix_sort"
    if "trampoline" in n: return "rocprim"
    return re.match(r"[A-Za-z0-9_]+", n).group(0) + g
end = 0
for r in rows[start_i:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    end = max(end, e)
    print("%8.3f ms  +%7.3f ms  q%-2s s%-2s %s  grid=%s" % (s / 1e6, (e - s) / 1e6, r["Queue_Id"], r["Stream_Id"], short(r["Kernel_Name"]), r["Grid_Size_X"]))
print("total span %.3f ms" % (end / 1e6))
