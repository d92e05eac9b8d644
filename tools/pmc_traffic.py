#!/usr/bin/env python3
"""Merges into profiles/pmc_traffic.json the per-kernel HBM traffic of two rocprofv3 --pmc run directories (FETCH_SIZE and WRITE_SIZE
collected in separate passes, as MI355X_MICROARCH.md prescribes), keyed by the per-GPU problem size the runs used:
    pmc_traffic.py <fetch_dir> <write_dir> profiles/pmc_traffic.json <log_n> [label]
-> json[kernel]["by_log_n"][log_n] (bench.py reports `roofline.traffic` only for the size it is running).  HBM bytes per launch = (f * FETCH_SIZE + WRITE_SIZE) * 1024 with the
gfx950 correction f calibrated PER ACCESS PATTERN (tools/gather_calib.hip, profiles/r03_gather_calib.json: known 4 GiB per launch): f = 2 for wide coalesced streams (the guide's
case: FETCH_SIZE counts a 128-byte request as 64 bytes), f = 2 for 128-byte random gathers (k_accumulate<G2>) and for 64-byte records read by consecutive lanes, but f = 1 for
64-byte gathers at random indices (k_accumulate<G1>'s window-table reads: FETCH_SIZE = 0.9997 of the known bytes).  The uncorrected figure is kept too."""
import collections, csv, glob, json, re, sys

NAMES = {"k_accumulate<G1>": ["msm_accumulate_g1"], "k_accumulate<G2>": ["msm_accumulate_g2"], "k_accumulate_pf<G1>": ["msm_accumulate_g1"], "k_accumulate_pf<G2>": ["msm_accumulate_g2"],
         "k_rs_scatter": ["msm_sort_pass"], "radix_sort_onesweep": ["msm_radix_sort(rocprim)"],
         "k_reduce_l1<G1>": ["msm_reduce_l1"], "k_reduce_wave<G1>": ["msm_reduce_wave"], "k_ntt_pass29": ["ntt_pass_contig", "ntt_pass_strided"],
         "k_ntt_pass": ["ntt_pass_contig", "ntt_pass_strided"], "k_msm_digits": ["msm_digits"]}


# FETCH_SIZE correction by access pattern (profiles/r03_gather_calib.json); everything else: wide streams, 2
FETCH_FACTOR = {"k_accumulate<G1>": 1.0, "k_accumulate_pf<G1>": 1.0}


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, disp = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = re.sub(r"void |zkmi::|rocprim::ROCPRIM_\d+_NS::|detail::", "", r["Kernel_Name"])
        g = "<G2>" if "Fp2" in n else ("<G1>" if "FpParams" in n else "")
        k = re.match(r"[A-Za-z0-9_]+", n).group(0)
        k = k + g if (k + g) in NAMES else k
        tot[k] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return {k: tot[k] / len(disp[k]) for k in tot}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --steps 2 --warmup 1 --no-cpu-baseline; hbm_bytes = "
        "(f*FETCH_SIZE + WRITE_SIZE)*1024, f calibrated per access pattern on known byte counts (tools/gather_calib.hip): 2 for streams and 128-B gathers, "
        "1 for k_accumulate<G1>'s random 64-B gathers; the uncorrected figure is given too")
log_n = sys.argv[4]
label = sys.argv[5] if len(sys.argv) > 5 else ""
try:
    out = json.load(open(sys.argv[3]))
except Exception:
    out = {}
out = {k: v for k, v in out.items() if isinstance(v, dict) and "by_log_n" in v}  # drop entries of the old un-keyed format
shown = {}
for k, names in NAMES.items():
    if k in fetch and k in write:
        for nm in names:
            f = FETCH_FACTOR.get(k, 2.0)
            e = {"FETCH_SIZE_KB_avg": fetch[k], "WRITE_SIZE_KB_avg": write[k], "hbm_bytes_per_launch": int((f * fetch[k] + write[k]) * 1024), "fetch_correction_factor": f,
                 "hbm_bytes_per_launch_uncorrected": int((fetch[k] + write[k]) * 1024), "rocprof_kernel": k, "taken_at": label}
            out.setdefault(nm, {"by_log_n": {}})["by_log_n"][log_n] = e
            shown[nm] = e["hbm_bytes_per_launch"]
out["_note"] = note
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(shown)
