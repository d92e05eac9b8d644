#!/bin/bash
# Round 5, batch Z (evidence at HEAD): the whole -m gpu suite, smoke, the default bench line (export blocks now before the big-memory blocks), rocprofv3 kernel stats of the
# 2^20 block
set -u
O=gpurun_out/${1:-rnd5z}
mkdir -p $O
export TMPDIR=/tmp
( time timeout 1700 python -m pytest tests -m gpu -q --durations=12 ) > $O/pytest_full.txt 2>&1; echo "rc=$?" >> $O/pytest_full.txt; tail -22 $O/pytest_full.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 1500 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<PY
import json
d=json.loads([l for l in open("$O/bench_default_line.json") if l.startswith("{")][-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["traffic"])
print("host", d["prove_ms_host_inputs"]["value"], "witness", d["prove_ms_witness_like_scalars"]["value"], "2p24", d["at_2p24"]["prove_ms"], d["at_2p24"].get("prove_ms_witness_like_scalars"))
print([ (k, d[k]["prove_ms"], d[k]["roofline"]["traffic"]) for k in d if k.startswith("plonk_2p")])
e=d.get("export_path_groth16",{}); print("g16 export", {k:e.get(k) for k in ("cold_ProveWithPK_ms","Preprocess_ms","warm_ProveWithPK_ms","zk_bn254_groth16_prove_r1cs_ms","warm_over_prove","ok","error")})
e=d.get("export_path",{}); print("plonk export", {k:e.get(k) for k in ("warm_PlonkProveWithPK_ms","warm_over_prove","ok","error")}, e.get("prove_process",{}).get("cold_PlonkProveWithPK_ms"))
print("parity_error", d.get("parity_error"), "cpu", d.get("cpu_baseline",{}).get("prove_ms"))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/$O/prof_2p20 -- python3 bench.py --steps 100 --no-2p24 --no-plonk --no-micro --no-export --no-cpu-baseline --no-host-inputs > $O/bench_2p20_under_rocprof.json 2> $O/rocprof_2p20.err
python tools/summarize_rocprof.py $O/prof_2p20 "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 (2^20 block only; 100 timed + 100 profiled-pass proofs + warm-up), round 5 HEAD" > $O/bench_2p20_kernel_stats.md 2>> $O/rocprof_2p20.err
rm -rf $O/prof_2p20
head -12 $O/bench_2p20_kernel_stats.md
