#!/bin/bash
# Round 5, batch AA: the G2 subgroup test as two kernels ([x0]P under a two-waves register bound, then the psi tail): key tests, then the kernels inside cold ProveWithPK calls
set -u
O=gpurun_out/${1:-rnd5aa}
mkdir -p $O /tmp/g16
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_keyio.py tests/test_gpu_goffi.py tests/test_gpu_plonk.py -m gpu -x -q > $O/pytest_sel.txt 2>&1; echo "rc=$?" >> $O/pytest_sel.txt; tail -3 $O/pytest_sel.txt
python tools/export_bench_groth16.py make /tmp/g16 20 > $O/g16_make.json 2> $O/g16_make.err
timeout 900 python tools/export_bench_groth16.py preprocess /tmp/g16 > $O/g16_preprocess.json 2> $O/g16_preprocess.err; echo "preprocess rc=$?"
for rep in 1 2 3 4; do
timeout 900 python tools/export_bench_groth16.py prove /tmp/g16 4 > $O/g16_prove_$rep.json 2> $O/g16_prove.err; python - <<PY
import json
d=json.load(open("$O/g16_prove_$rep.json")); p=d["cold_phases"]; k=d["cold_largest_kernels_ms"]
print("g16 cold", d["cold_ProveWithPK_ms"], "hip_init", p.get("hip_init"), "pk_read", p.get("pk_read"), "= g2 part", p.get("pk_read_upload_g2_part"), "rest", p.get("pk_read_upload_rest"), "wait", p.get("pk_read_decompress_wait"), "load", p.get("pk_read_load"), "| kernels g2 sqrt", k.get("g2_decompress"), "x0_mul", k.get("g2_x0_mul"), "tail", k.get("g2_subgroup"), "g1", k.get("g1_decompress"), "| warm", d["warm_ProveWithPK_ms"], "verifies", d["verifies"])
PY
done
