#!/bin/bash
# Round 5, batch Q: big uploads through the pinned ring (ctx.hip h2d_big): the new r1cs test, key / export tests, then the cold calls of both export workers
set -u
O=gpurun_out/${1:-rnd5q}
mkdir -p $O /tmp/g16 /tmp/plk
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "r1cs" > $O/pytest_r1cs.txt 2>&1; echo "rc=$?" >> $O/pytest_r1cs.txt; tail -3 $O/pytest_r1cs.txt
timeout 1500 python -m pytest tests/test_gpu_keyio.py tests/test_gpu_goffi.py tests/test_gpu_plonk.py -m gpu -x -q > $O/pytest_sel.txt 2>&1; echo "rc=$?" >> $O/pytest_sel.txt; tail -3 $O/pytest_sel.txt
python tools/export_bench_groth16.py make /tmp/g16 20 > $O/g16_make.json 2> $O/g16_make.err
for rep in 1 2 3; do
timeout 900 python tools/export_bench_groth16.py preprocess /tmp/g16 > $O/g16_preprocess_$rep.json 2> $O/g16_preprocess.err; echo "preprocess rc=$?"
timeout 900 python tools/export_bench_groth16.py prove /tmp/g16 10 > $O/g16_prove_$rep.json 2> $O/g16_prove.err; echo "prove rc=$?"; python - <<PY
import json
f=lambda ph: {k:v for k,v in ph.items() if v>=3}
d=json.load(open("$O/g16_preprocess_$rep.json")); print("Preprocess", d["Preprocess_ms"], f(d["phases"]), "verifies", d["verifies"])
d=json.load(open("$O/g16_prove_$rep.json")); print("cold prove", d["cold_ProveWithPK_ms"], f(d["cold_phases"]), "second", d["second_ProveWithPK_ms"], "warm", d["warm_ProveWithPK_ms"], "verifies", d["verifies"], d["warm_proof_verifies"])
PY
done
python tools/export_bench.py make /tmp/plk > $O/plk_make.json 2> $O/plk_make.err
for rep in 1 2; do
timeout 900 python tools/export_bench.py preprocess /tmp/plk > $O/plk_preprocess_$rep.json 2> $O/plk_preprocess.err; echo "plonk preprocess rc=$?"
timeout 900 python tools/export_bench.py prove /tmp/plk 10 > $O/plk_prove_$rep.json 2> $O/plk_prove.err; echo "plonk prove rc=$?"; python - <<PY
import json
f=lambda ph: {k:v for k,v in ph.items() if v>=3}
d=json.load(open("$O/plk_prove_$rep.json")); print("PLONK cold", d["cold_PlonkProveWithPK_ms"], f(d["cold_phases"]), "warm", d["warm_PlonkProveWithPK_ms"], "verifies", d["verifies"])
PY
done
