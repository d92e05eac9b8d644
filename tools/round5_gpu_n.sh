#!/bin/bash
# Round 5, batch N: the warm export calls prove while the texts are still being compared (ZKMI_EXPORT_SPECULATE, default on): key / export / PLONK tests,
# then the Groth16 and PLONK export workers with the switch on and off.
set -u
O=gpurun_out/${1:-rnd5n}
mkdir -p $O /tmp/g16 /tmp/plk
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_keyio.py tests/test_gpu_goffi.py tests/test_gpu_plonk.py -m gpu -x -q > $O/pytest_sel.txt 2>&1; echo "rc=$?" >> $O/pytest_sel.txt; tail -4 $O/pytest_sel.txt
python tools/export_bench_groth16.py make /tmp/g16 20 > $O/g16_make.json 2> $O/g16_make.err
timeout 900 python tools/export_bench_groth16.py preprocess /tmp/g16 > $O/g16_preprocess.json 2> $O/g16_preprocess.err; echo "preprocess rc=$?"
for sp in 1 0 1 0; do
ZKMI_EXPORT_SPECULATE=$sp timeout 900 python tools/export_bench_groth16.py prove /tmp/g16 20 > $O/g16_prove_spec$sp.json 2> $O/g16_prove.err; echo "prove rc=$?"; python - <<PY
import json
d=json.load(open("$O/g16_prove_spec$sp.json")); w=d["warm_phases_per_call"]; print("speculate=$sp cold", d["cold_ProveWithPK_ms"], "warm", d["warm_ProveWithPK_ms"], round(d["warm_ProveWithPK_ms"]/(w["groth16_prove"]+w["r1cs_solve_abc"]),3), "verifies", d["verifies"], d["warm_proof_verifies"], d["wrong_public_input_rejected"], w)
PY
done
python tools/export_bench.py make /tmp/plk > $O/plk_make.json 2> $O/plk_make.err
timeout 900 python tools/export_bench.py preprocess /tmp/plk > $O/plk_preprocess.json 2> $O/plk_preprocess.err; echo "plonk preprocess rc=$?"
for sp in 1 0 1 0; do
ZKMI_EXPORT_SPECULATE=$sp timeout 900 python tools/export_bench.py prove /tmp/plk 20 > $O/plk_prove_spec$sp.json 2> $O/plk_prove.err; echo "plonk prove rc=$?"; python - <<PY
import json
d=json.load(open("$O/plk_prove_spec$sp.json")); w=d["warm_phases_per_call"]; print("speculate=$sp cold", d["cold_PlonkProveWithPK_ms"], "warm", d["warm_PlonkProveWithPK_ms"], d.get("warm_over_prove"), "verifies", d.get("verifies"), d.get("warm_proof_verifies"), w)
PY
done
