#!/usr/bin/env python3
"""A/B driver for the experiment switches (ZK_EXP in csrc/ctx.hpp): runs bench.py once per configuration against libzkmi_exp.so -- the build that reads
the switches; the shipped libzkmi.so has them compiled out -- and prints one compact JSON line per configuration: Groth16 2^20 / 2^24, PLONK 2^22,
the 2^26 MSM, and the per-kernel milliseconds that matter for the schedule (sort, accumulate, tails, transforms).

    python tools/ab_bench.py OUT.jsonl  "name:VAR=1,VAR2=0"  "other:VAR=2" ...        (a bare "base" runs the defaults)
    options: --steps N   --only 2p20|2p24|plonk|micro (repeatable; default all four)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    argv, args = sys.argv[1:], []
    steps, only = 60, []
    while argv:
        if argv[0] == "--steps":
            steps, argv = int(argv[1]), argv[2:]
        elif argv[0] == "--only":
            only.append(argv[1])
            argv = argv[2:]
        elif argv[0].startswith("--"):
            raise SystemExit("unknown option " + argv[0])
        else:
            args.append(argv[0])
            argv = argv[1:]
    out_path, cfgs = args[0], args[1:]
    only = only or ["2p20", "2p24", "plonk", "micro"]
    flags = ["--steps", str(steps), "--warmup", "5", "--no-cpu-baseline", "--no-host-inputs", "--no-export"]
    for k, f in (("2p24", "--no-2p24"), ("plonk", "--no-plonk"), ("micro", "--no-micro")):
        if k not in only:
            flags.append(f)
    with open(out_path, "a") as fo:
        for cfg in cfgs:
            name, _, kv = cfg.partition(":")
            env = dict(os.environ)
            lib, extra = ["--lib", "exp"], []
            for item in filter(None, kv.split(",")):
                k, _, v = item.partition("=")
                if k == "SCALARS":  # SCALARS=witness: the main instance's wire vector is witness-like (bench.py --scalars)
                    extra = ["--scalars", v]
                elif k == "LIB":  # LIB=product: the shipped libzkmi.so (no switches), e.g. against an experiments build made with other compile flags
                    lib = [] if v == "product" else ["--lib", "exp"]
                else:
                    env[k] = v
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + flags + lib + extra, capture_output=True, text=True, env=env, cwd=ROOT)
            lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not lines:
                rec = {"name": name, "env": kv, "error": (p.stderr or p.stdout)[-600:]}
            else:
                d = json.loads(lines[-1])
                km = d["roofline"]["kernel_ms_per_step"]
                rec = {"name": name, "env": kv, "prove_2p20_ms": d["ms_per_step"], "proof_sha": d["proof_sha"], "parity_error": d.get("parity_error"),
                       "kernels_2p20": {k: km[k] for k in list(km)[:12]},
                       "valu_frac": (d["roofline"].get("valu") or {}).get("frac")}
                if "at_2p24" in d:
                    rec["prove_2p24_ms"] = d["at_2p24"]["prove_ms"]
                    rec["ok_2p24"] = d["at_2p24"]["proof_bytes_match_recombination"]
                for k in d:
                    if k.startswith("plonk_2p"):
                        rec["plonk_ms"] = d[k]["prove_ms"]
                        rec["plonk_ok"] = d[k]["proof_verifies"]
                        rec["plonk_kernels"] = d[k]["kernel_ms_per_proof"]
                if "micro_2p26" in d:
                    m = d["micro_2p26"]
                    rec.update(msm_2p26_ms=m["g1_msm_ms"], msm_2p26_tables_ms=m["g1_msm_window_tables_ms"], ntt_2p26_ms=m["ntt_ms"],
                               micro_ok=bool(m["equals_split_recombination"] and m["equals_window_table_path"]))
            fo.write(json.dumps(rec) + "\n")
            fo.flush()
            print(json.dumps({k: v for k, v in rec.items() if not isinstance(v, dict)}), flush=True)


if __name__ == "__main__":
    main()
