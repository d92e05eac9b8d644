#!/bin/bash
# Round 5, batch Y (experiment): every translation unit's code object loaded by an empty kernel while the start-up's streams are being created (1), after them (2), or
# at each unit's first real launch as before (0) -- cold calls of both export workers
set -u
O=gpurun_out/${1:-rnd5y}
mkdir -p $O /tmp/g16 /tmp/plk
export TMPDIR=/tmp
python tools/export_bench_groth16.py make /tmp/g16 20 > $O/g16_make.json 2> $O/g16_make.err
timeout 900 python tools/export_bench_groth16.py preprocess /tmp/g16 > $O/g16_preprocess.json 2> $O/g16_preprocess.err; echo "preprocess rc=$?"
python tools/export_bench.py make /tmp/plk > $O/plk_make.json 2> $O/plk_make.err
timeout 900 python tools/export_bench.py preprocess /tmp/plk > $O/plk_preprocess.json 2> $O/plk_preprocess.err; echo "plonk preprocess rc=$?"
for rep in 1 2 3 4 5; do
for m in 0 1 2; do
ZKMI_TMP_WARM_UNITS=$m timeout 900 python tools/export_bench_groth16.py prove /tmp/g16 4 > $O/g16_prove_${m}_$rep.json 2> $O/g16_prove.err; python - <<PY
import json
d=json.load(open("$O/g16_prove_${m}_$rep.json")); p=d["cold_phases"]
print("mode $m: g16 cold", d["cold_ProveWithPK_ms"], "hip_init", p.get("hip_init"), "warm_units", p.get("warm_units"), "pk_read", p.get("pk_read"), "circuit", p.get("circuit_to_device"), "prove", p.get("groth16_prove"), "init+key+prove", round(p.get("hip_init",0)+p.get("pk_read",0)+p.get("groth16_prove",0),1), "| warm", d["warm_ProveWithPK_ms"], "verifies", d["verifies"])
PY
ZKMI_TMP_WARM_UNITS=$m timeout 900 python tools/export_bench.py prove /tmp/plk 4 > $O/plk_prove_${m}_$rep.json 2> $O/plk_prove.err; python - <<PY
import json
d=json.load(open("$O/plk_prove_${m}_$rep.json")); p=d["cold_phases"]
print("mode $m: plonk cold", d["cold_PlonkProveWithPK_ms"], "hip_init", p.get("hip_init"), "warm_units", p.get("warm_units"), "srs_decode", p.get("srs_decode"), "pk_resident", p.get("pk_resident"), "prove", p.get("plonk_prove"), "init+srs+pk+prove", round(p.get("hip_init",0)+p.get("srs_decode",0)+p.get("pk_resident",0)+p.get("plonk_prove",0),1), "| warm", d["warm_PlonkProveWithPK_ms"], "verifies", d["verifies"])
PY
done
done
