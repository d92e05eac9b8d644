#!/bin/bash
# Round 5, batch W: the whole GPU suite and the default bench line at HEAD (after the cold-path and warm-call work)
set -u
O=gpurun_out/${1:-rnd5w}
mkdir -p $O
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 1500 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<PY
import json
d=json.loads([l for l in open("$O/bench_default_line.json") if l.startswith("{")][-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["traffic"])
print("host", d["prove_ms_host_inputs"]["value"], "witness", d["prove_ms_witness_like_scalars"]["value"], "2p24", d["at_2p24"]["prove_ms"], d["at_2p24"].get("prove_ms_witness_like_scalars"))
print([ (k, d[k]["prove_ms"], d[k]["roofline"]["traffic"]) for k in d if k.startswith("plonk_2p")])
e=d.get("export_path_groth16",{}); print("g16 export", {k:e.get(k) for k in ("cold_ProveWithPK_ms","Preprocess_ms","warm_ProveWithPK_ms","zk_bn254_groth16_prove_r1cs_ms","warm_over_prove","ok","error")})
e=d.get("export_path",{}); print("plonk export", {k:e.get(k) for k in ("warm_PlonkProveWithPK_ms","warm_over_prove","ok","error")})
print("parity_error", d.get("parity_error"))
PY
