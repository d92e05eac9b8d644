#!/bin/bash
# Round 5, batch V (experiment): the G2 subgroup test as its own kernel -- on the 29-bit multiplier (0) or on the saturated one (2)
set -u
O=gpurun_out/${1:-rnd5v}
mkdir -p $O /tmp/g16
export TMPDIR=/tmp
python tools/export_bench_groth16.py make /tmp/g16 20 > $O/g16_make.json 2> $O/g16_make.err
timeout 900 python tools/export_bench_groth16.py preprocess /tmp/g16 > $O/g16_preprocess.json 2> $O/g16_preprocess.err; echo "preprocess rc=$?"
for rep in 1 2 3; do
for v in 0 2; do
ZKMI_TMP_G2SUB=$v timeout 900 python tools/export_bench_groth16.py prove /tmp/g16 2 > $O/g16_prove_${v}_$rep.json 2> $O/g16_prove.err; python - <<PY
import json
d=json.load(open("$O/g16_prove_${v}_$rep.json")); p=d["cold_phases"]; k=d["cold_largest_kernels_ms"]
print("variant $v: g16 cold", d["cold_ProveWithPK_ms"], "hip_init", p.get("hip_init"), "pk_read", p.get("pk_read"), "= g2 part", p.get("pk_read_upload_g2_part"), "rest", p.get("pk_read_upload_rest"), "wait", p.get("pk_read_decompress_wait"), "load", p.get("pk_read_load"), "| kernels g2 sqrt", k.get("g2_decompress"), "subgroup", k.get("g2_subgroup"), "g1", k.get("g1_decompress"), "verifies", d["verifies"])
PY
done
done
