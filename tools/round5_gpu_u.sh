#!/bin/bash
# Round 5, batch U (experiment): which uploads should take the pinned ring -- all from 1 MB, only those from 64 MB, none -- in the cold calls of both export
# workers; also the first run of the square roots and the G2 subgroup test on the 29-bit multiplier (key tests first)
set -u
O=gpurun_out/${1:-rnd5u}
mkdir -p $O /tmp/g16 /tmp/plk
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_keyio.py -m gpu -x -q > $O/pytest_keyio.txt 2>&1; echo "rc=$?" >> $O/pytest_keyio.txt; tail -3 $O/pytest_keyio.txt
python tools/export_bench_groth16.py make /tmp/g16 20 > $O/g16_make.json 2> $O/g16_make.err
timeout 900 python tools/export_bench_groth16.py preprocess /tmp/g16 > $O/g16_preprocess.json 2> $O/g16_preprocess.err; echo "preprocess rc=$?"
python tools/export_bench.py make /tmp/plk > $O/plk_make.json 2> $O/plk_make.err
timeout 900 python tools/export_bench.py preprocess /tmp/plk > $O/plk_preprocess.json 2> $O/plk_preprocess.err; echo "plonk preprocess rc=$?"
for rep in 1 2 3; do
for mb in 1 64 100000; do
ZKMI_TMP_H2D_MIN_MB=$mb timeout 900 python tools/export_bench_groth16.py prove /tmp/g16 2 > $O/g16_prove_${mb}_$rep.json 2> $O/g16_prove.err; python - <<PY
import json
d=json.load(open("$O/g16_prove_${mb}_$rep.json")); p=d["cold_phases"]; k=d["cold_largest_kernels_ms"]
print("ring from $mb MB: g16 cold", d["cold_ProveWithPK_ms"], "hip_init", p.get("hip_init"), "pk_read", p.get("pk_read"), "= g2 part", p.get("pk_read_upload_g2_part"), "rest", p.get("pk_read_upload_rest"), "wait", p.get("pk_read_decompress_wait"), "load", p.get("pk_read_load"), "| circuit", p.get("circuit_to_device"), "r1cs", p.get("circuit_r1cs_load"), "order", p.get("circuit_order_uploads"), "| kernels g2", k.get("g2_decompress"), k.get("g2_subgroup"), "g1", k.get("g1_decompress"), "verifies", d["verifies"])
PY
ZKMI_TMP_H2D_MIN_MB=$mb timeout 900 python tools/export_bench.py prove /tmp/plk 2 > $O/plk_prove_${mb}_$rep.json 2> $O/plk_prove.err; python - <<PY
import json
d=json.load(open("$O/plk_prove_${mb}_$rep.json")); p=d["cold_phases"]
print("ring from $mb MB: plonk cold", d["cold_PlonkProveWithPK_ms"], "hip_init", p.get("hip_init"), "srs_decode", p.get("srs_decode"), "pk_text_to_device", p.get("pk_text_to_device"), "pk_resident", p.get("pk_resident"), "verifies", d["verifies"])
PY
done
done
