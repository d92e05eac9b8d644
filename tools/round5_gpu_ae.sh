#!/bin/bash
# Round 5, batch AE (experiment): runtime environment knobs at a cold call's start -- HSA_ENABLE_SDMA=0 (no SDMA queues: copies as blit kernels), default
set -u
O=gpurun_out/${1:-rnd5ae}
mkdir -p $O /tmp/g16 /tmp/plk
export TMPDIR=/tmp
python tools/export_bench_groth16.py make /tmp/g16 20 > $O/g16_make.json 2> $O/g16_make.err
timeout 900 python tools/export_bench_groth16.py preprocess /tmp/g16 > $O/g16_preprocess.json 2> $O/g16_preprocess.err; echo "preprocess rc=$?"
python tools/export_bench.py make /tmp/plk > $O/plk_make.json 2> $O/plk_make.err
timeout 900 python tools/export_bench.py preprocess /tmp/plk > $O/plk_preprocess.json 2> $O/plk_preprocess.err; echo "plonk preprocess rc=$?"
for rep in 1 2 3 4; do
for v in default sdma0; do
if [ $v = sdma0 ]; then export HSA_ENABLE_SDMA=0; else unset HSA_ENABLE_SDMA; fi
timeout 900 python tools/export_bench_groth16.py prove /tmp/g16 6 > $O/g16_prove_${v}_$rep.json 2> $O/g16_prove.err; python - <<PY
import json
d=json.load(open("$O/g16_prove_${v}_$rep.json")); p=d["cold_phases"]
print("$v: g16 cold", d["cold_ProveWithPK_ms"], "hip_init", p.get("hip_init"), "pk_read", p.get("pk_read"), "circuit", p.get("circuit_to_device"), "prove", p.get("groth16_prove"), "| second", d["second_ProveWithPK_ms"], "| warm", d["warm_ProveWithPK_ms"], "verifies", d["verifies"])
PY
timeout 900 python tools/export_bench.py prove /tmp/plk 6 > $O/plk_prove_${v}_$rep.json 2> $O/plk_prove.err; python - <<PY
import json
d=json.load(open("$O/plk_prove_${v}_$rep.json")); p=d["cold_phases"]
print("$v: plonk cold", d["cold_PlonkProveWithPK_ms"], "hip_init", p.get("hip_init"), "srs_decode", p.get("srs_decode"), "pk_resident", p.get("pk_resident"), "prove", p.get("plonk_prove"), "| second", d["second_PlonkProveWithPK_ms"], "| warm", d["warm_PlonkProveWithPK_ms"], "verifies", d["verifies"])
PY
done
done
