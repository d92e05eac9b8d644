#!/usr/bin/env python3
"""Kernel timeline of ONE proof out of a rocprofv3 `--kernel-trace --output-format csv` run of bench.py: start offset, duration, hardware queue, kernel.
    timeline_proof.py <dir or kernel_trace.csv[.gz]> groth16|groth16_2p24|plonk [min_us=100]  > profiles/<name>.txt
A proof is located by its anchor kernel (for Groth16 the last pass of computeH's closing transform, the `true` variant of k_ntt_pass29 -- grid 2^18 or 2^22
lanes; k_h_final in traces older than that fusion --, k_quotient for PLONK): the period is the distance between two
consecutive anchors, the proof starts after the longest idle gap in the period before the anchor.  The summary lines give the span, the time at least one
kernel was running, and per kernel the sum of durations inside the proof."""
import collections, csv, glob, gzip, io, os, re, sys


def short(n):
    n = re.sub(r"void |zkmi::|rocprim::ROCPRIM_\d+_NS::|detail::", "", n)
    g = "<G2>" if "Fp2" in n else ("<G1>" if "FpParams" in n else "")
    if closing(n):
        g = "(closing)"
    return re.match(r"[A-Za-z0-9_]+", n).group(0) + g


def closing(name):
    return "k_ntt_pass29<" in name and "true>" in name


def main():
    src, what = sys.argv[1], sys.argv[2]
    min_ns = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 100e3
    if os.path.isdir(src):
        src = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
    fh = io.TextIOWrapper(gzip.open(src)) if src.endswith(".gz") else open(src)
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"], r["Grid_Size_X"]) for r in csv.DictReader(fh)]
    rows.sort()
    if what == "plonk":
        anchors = [r[0] for r in rows if "k_quotient" in r[3]]
    else:
        grid = "16777216" if what == "groth16_2p24" else "1048576"
        anchors = [r[0] for r in rows if "k_h_final" in r[3] and r[4] == grid]
        if not anchors:
            grid = "4194304" if what == "groth16_2p24" else "262144"
            anchors = [r[0] for r in rows if closing(r[3]) and r[4] == grid]
    if len(anchors) < 3:
        raise SystemExit("fewer than three %s proofs in the trace" % what)
    a0, a1 = anchors[-2], anchors[-1]
    period = a1 - a0
    # the longest idle gap in [a0 - period, a0]
    cur_end, best_gap, start = None, -1, a0 - period
    for s, e, _, _, _ in rows:
        if e < a0 - period or s > a0:
            continue
        if cur_end is not None and s - cur_end > best_gap:
            best_gap, start = s - cur_end, s
        cur_end = max(cur_end or e, e)
    sel = [r for r in rows if start <= r[0] < start + period]
    print("# one %s proof of the trace: %d kernels, period %.3f ms (anchor to anchor); kernels of at least %.0f us listed" % (what, len(sel), period / 1e6, min_ns / 1e3))
    busy, ce = 0, None
    tot = collections.defaultdict(lambda: [0, 0])
    for s, e, q, name, grid in sel:
        k = short(name)
        tot[k][0] += 1
        tot[k][1] += e - s
        if ce is None or s > ce:
            busy += e - s
            ce = e
        elif e > ce:
            busy += e - ce
            ce = e
        if e - s >= min_ns:
            print("%8.3f ms  +%7.3f ms  q%-2s %-26s grid=%s" % ((s - start) / 1e6, (e - s) / 1e6, q, k, grid))
    end = max(r[1] for r in sel)
    print("# span %.3f ms, some kernel running for %.3f ms of it" % ((end - start) / 1e6, busy / 1e6))
    print("# sum of kernel durations inside the proof (kernels overlap, so the sums exceed the span):")
    for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:24]:
        print("#   %-28s x%-4d %8.3f ms" % (k, c, t / 1e6))


if __name__ == "__main__":
    main()
