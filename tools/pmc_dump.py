"""Per-dispatch counter values of the kernels whose name matches a pattern, from a rocprofv3 --pmc run directory (counter_collection.csv): one JSON line per
dispatch with its grid -- what tells PLONK's batched accumulate launch (three bucket sets: grid y = 3) from the single-vector ones.
    python tools/pmc_dump.py <dir> <counter> <name-regex>  > dispatches.jsonl"""
import csv
import glob
import json
import re
import sys

d, counter, pat = sys.argv[1], sys.argv[2], re.compile(sys.argv[3])
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
with open(f) as fh:
    rd = csv.DictReader(fh)
    cols = rd.fieldnames
    grid = [c for c in cols if c.lower().startswith("grid")]
    wg = [c for c in cols if c.lower().startswith("workgroup")]
    print(json.dumps({"columns": cols}))
    for r in rd:
        if r["Counter_Name"] != counter or not pat.search(r["Kernel_Name"]):
            continue
        g2 = "<G2>" if "Fp2" in r["Kernel_Name"] else "<G1>" if "FpParams" in r["Kernel_Name"] else ""
        print(json.dumps({"dispatch": r["Dispatch_Id"], "kernel": re.match(r"[A-Za-z0-9_: ]+", r["Kernel_Name"].replace("void zkmi::", "")).group(0).strip() + g2,
                          "grid": [r[c] for c in grid], "workgroup": [r[c] for c in wg], "value": float(r["Counter_Value"])}))
