#!/bin/bash
# Round 5, batch L (HEAD): key / export / multi-device tests, the Groth16 export worker (values uploaded while the content keys are compared), the default bench line.
set -u
O=gpurun_out/${1:-rnd5l}
mkdir -p $O /tmp/g16
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_keyio.py tests/test_gpu_goffi.py tests/test_gpu_multidev.py -m gpu -x -q > $O/pytest_sel.txt 2>&1; echo "rc=$?" >> $O/pytest_sel.txt; tail -4 $O/pytest_sel.txt
python tools/export_bench_groth16.py make /tmp/g16 20 > $O/g16_make.json 2> $O/g16_make.err
timeout 900 python tools/export_bench_groth16.py preprocess /tmp/g16 > $O/g16_preprocess.json 2> $O/g16_preprocess.err; echo "preprocess rc=$?"
for rep in 1 2 3; do
timeout 900 python tools/export_bench_groth16.py prove /tmp/g16 20 > $O/g16_prove_$rep.json 2> $O/g16_prove.err; echo "prove rc=$?"; python - <<PY
import json
d=json.load(open("$O/g16_prove_$rep.json")); w=d["warm_phases_per_call"]; print("cold", d["cold_ProveWithPK_ms"], "warm", d["warm_ProveWithPK_ms"], round(d["warm_ProveWithPK_ms"]/(w["groth16_prove"]+w["r1cs_solve_abc"]),3), w)
PY
done
timeout 1500 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<PY
import json
d=json.loads([l for l in open("$O/bench_default_line.json") if l.startswith("{")][-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["traffic"])
print("host", d["prove_ms_host_inputs"]["value"], "witness", d["prove_ms_witness_like_scalars"]["value"], "2p24", d["at_2p24"]["prove_ms"], d["at_2p24"].get("prove_ms_witness_like_scalars"))
print([ (k, d[k]["prove_ms"], d[k]["roofline"]["traffic"]) for k in d if k.startswith("plonk_2p")])
e=d.get("export_path_groth16",{}); print("g16 export", {k:e.get(k) for k in ("cold_ProveWithPK_ms","warm_ProveWithPK_ms","zk_bn254_groth16_prove_r1cs_ms","warm_over_prove","ok","error")})
e=d.get("export_path",{}); print("plonk export", {k:e.get(k) for k in ("warm_PlonkProveWithPK_ms","warm_over_prove","ok","error")})
print("parity_error", d.get("parity_error"))
PY
