#!/bin/bash
# Round 5, batch R: where the cold ProveWithPK's key / circuit phases go (finer laps), three fresh processes
set -u
O=gpurun_out/${1:-rnd5r}
mkdir -p $O /tmp/g16
export TMPDIR=/tmp
python tools/export_bench_groth16.py make /tmp/g16 20 > $O/g16_make.json 2> $O/g16_make.err
timeout 900 python tools/export_bench_groth16.py preprocess /tmp/g16 > $O/g16_preprocess.json 2> $O/g16_preprocess.err; echo "preprocess rc=$?"
for rep in 1 2 3; do
timeout 900 python tools/export_bench_groth16.py prove /tmp/g16 4 > $O/g16_prove_$rep.json 2> $O/g16_prove.err; echo "prove rc=$?"; python - <<PY
import json
f=lambda ph: {k:v for k,v in ph.items() if v>=1}
d=json.load(open("$O/g16_prove_$rep.json")); print("cold prove", d["cold_ProveWithPK_ms"], f(d["cold_phases"]), d["cold_largest_kernels_ms"])
PY
done
python - <<PY
import json
f=lambda ph: {k:v for k,v in ph.items() if v>=1}
d=json.load(open("$O/g16_preprocess.json")); print("Preprocess", d["Preprocess_ms"], f(d["phases"]), d["largest_kernels_ms"])
PY
