#!/bin/bash
# Round 5, batch AF: key texts handed back through the pinned ring (ctx.hip d2h_big): key / export / PLONK tests, then Preprocess of both export workers
set -u
O=gpurun_out/${1:-rnd5af}
mkdir -p $O /tmp/g16 /tmp/plk
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_keyio.py tests/test_gpu_goffi.py tests/test_gpu_plonk.py -m gpu -x -q > $O/pytest_sel.txt 2>&1; echo "rc=$?" >> $O/pytest_sel.txt; tail -3 $O/pytest_sel.txt
python tools/export_bench_groth16.py make /tmp/g16 20 > $O/g16_make.json 2> $O/g16_make.err
python tools/export_bench.py make /tmp/plk > $O/plk_make.json 2> $O/plk_make.err
for rep in 1 2 3 4; do
timeout 900 python tools/export_bench_groth16.py preprocess /tmp/g16 > $O/g16_preprocess_$rep.json 2> $O/g16_preprocess.err; python - <<PY
import json
d=json.load(open("$O/g16_preprocess_$rep.json")); p=d["phases"]
print("g16 Preprocess", d["Preprocess_ms"], "hip_init", p.get("hip_init"), "circuit", p.get("circuit_to_device"), "setup", p.get("groth16_setup"), "pk_write_hex", p.get("pk_write_hex"), "verifies", d["verifies"])
PY
timeout 900 python tools/export_bench.py preprocess /tmp/plk > $O/plk_preprocess_$rep.json 2> $O/plk_preprocess.err; python - <<PY
import json
d=json.load(open("$O/plk_preprocess_$rep.json")); p=d["phases"]
print("plonk Preprocess", d["PlonkPreprocess_ms"], "hip_init", p.get("hip_init"), "setup", p.get("plonk_setup"), "pk_write_hex", p.get("pk_write_hex"), "srs_save", p.get("srs_save"), "verifies", d["verifies"])
PY
done
