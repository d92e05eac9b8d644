#!/bin/bash
# Round 5, batch S (experiment): the key's decompression stream in a cold ProveWithPK -- second ordinary stream (0), a high-priority stream (1), the upload stream (2)
set -u
O=gpurun_out/${1:-rnd5s}
mkdir -p $O /tmp/g16
export TMPDIR=/tmp
python tools/export_bench_groth16.py make /tmp/g16 20 > $O/g16_make.json 2> $O/g16_make.err
timeout 900 python tools/export_bench_groth16.py preprocess /tmp/g16 > $O/g16_preprocess.json 2> $O/g16_preprocess.err; echo "preprocess rc=$?"
for rep in 1 2 3; do
for mode in 0 1 2; do
ZKMI_TMP_PKREAD_MODE=$mode timeout 900 python tools/export_bench_groth16.py prove /tmp/g16 2 > $O/g16_prove_${mode}_$rep.json 2> $O/g16_prove.err; python - <<PY
import json
d=json.load(open("$O/g16_prove_${mode}_$rep.json")); p=d["cold_phases"]; k=d["cold_largest_kernels_ms"]
print("mode $mode cold", d["cold_ProveWithPK_ms"], "hip_init", p.get("hip_init"), "pk_read", p.get("pk_read"), "= g2 up", p.get("pk_read_upload_g2_h2d"), "rest", p.get("pk_read_upload_rest"), "wait", p.get("pk_read_decompress_wait"), "load", p.get("pk_read_load"), "tmp", p.get("pk_read_tmp_stream"), "| circuit", p.get("circuit_to_device"), "r1cs", p.get("circuit_r1cs_load"), "order", p.get("circuit_order_uploads"), "| kernels g2", k.get("g2_decompress"), "g1", k.get("g1_decompress"), "verifies", d["verifies"])
PY
done
done
