#!/usr/bin/env python3
"""Condenses a rocprofv3 `--kernel-trace --stats --output-format csv` run directory into a small markdown table
(kept under profiles/).  usage: summarize_rocprof.py <dir-with-*_kernel_stats.csv> <title> > profiles/<name>.md"""
import csv
import glob
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"void |zkmi::|rocprim::ROCPRIM_\d+_NS::|detail::", "", name)
    m = re.match(r"([A-Za-z0-9_:]+)(<.*?>)?", name)
    base = m.group(1) if m else name
    if "Fp2" in name:
        base += "<G2>"
    elif "FpParams" in name:
        base += "<G1>"
    elif "FrParams" in name and base.startswith("k_"):
        base += "<Fr>"
    if "onesweep" in name:
        base = "rocprim::radix_sort_onesweep"
    elif "trampoline_kernel" in name and "scan" in name:
        base = "rocprim::scan"
    elif "trampoline_kernel" in name:
        base = "rocprim::" + (re.search(r"wrapped_([a-z_]+)_config", name).group(1) if re.search(r"wrapped_([a-z_]+)_config", name) else "kernel")
    return base


def main():
    d, title = sys.argv[1], sys.argv[2]
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    rows = {}
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = short(r["Name"])
            calls, tot = int(r["Calls"]), int(r["TotalDurationNs"])
            c0, t0 = rows.get(k, (0, 0))
            rows[k] = (c0 + calls, t0 + tot)
    total = sum(t for _, t in rows.values())
    print("# %s\n" % title)
    print("source: `rocprofv3 --kernel-trace --stats --output-format csv` (%s)\n" % f.split("/")[-1])
    print("| kernel | calls | avg us | total ms | % |")
    print("|---|---:|---:|---:|---:|")
    for k, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print("| %s | %d | %.1f | %.3f | %.2f |" % (k, c, t / c / 1e3, t / 1e6, 100.0 * t / total))


if __name__ == "__main__":
    main()
