#!/bin/bash
# Round 5, batch I: lean start without high-priority streams until a second proof; pipelined transposes of the in-process multi-GPU computeH (virtual entries);
# cold calls of both export paths, three runs each.
set -u
O=gpurun_out/${1:-rnd5i}
mkdir -p $O /tmp/g16 /tmp/plk
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_keyio.py tests/test_gpu_goffi.py tests/test_gpu_multidev.py -m gpu -x -q --durations=5 > $O/pytest_sel.txt 2>&1; echo "rc=$?" >> $O/pytest_sel.txt; tail -12 $O/pytest_sel.txt
python tools/export_bench_groth16.py make /tmp/g16 20 > $O/g16_make.json 2> $O/g16_make.err
python tools/export_bench.py make /tmp/plk 19 > $O/plonk_make.json 2> $O/plonk_make.err
timeout 900 python tools/export_bench.py preprocess /tmp/plk > $O/plonk_preprocess.json 2> $O/plonk_preprocess.err; echo "plonk preprocess rc=$?"
for rep in 1 2 3; do
timeout 900 python tools/export_bench_groth16.py preprocess /tmp/g16 > $O/g16_preprocess_$rep.json 2> $O/g16_preprocess.err; echo "preprocess rc=$?"
timeout 900 python tools/export_bench_groth16.py prove /tmp/g16 10 > $O/g16_prove_$rep.json 2> $O/g16_prove.err; echo "prove rc=$?"
timeout 900 python tools/export_bench.py prove /tmp/plk 10 > $O/plonk_prove_$rep.json 2> $O/plonk_prove.err; echo "plonk prove rc=$?"
python - <<PY
import json
p=json.load(open("$O/g16_preprocess_$rep.json")); d=json.load(open("$O/g16_prove_$rep.json")); q=json.load(open("$O/plonk_prove_$rep.json"))
print("Preprocess", p["Preprocess_ms"], {k:x for k,x in p["phases"].items() if x>8})
print("cold prove", d["cold_ProveWithPK_ms"], {k:x for k,x in d["cold_phases"].items() if x>3}, "second", d["second_ProveWithPK_ms"], d["second_phases"], "warm", d["warm_ProveWithPK_ms"], d["warm_phases_per_call"].get("groth16_prove"))
print("PLONK cold", q["cold_PlonkProveWithPK_ms"], {k:x for k,x in q["cold_phases"].items() if x>3}, "second", q["second_PlonkProveWithPK_ms"], "warm", q["warm_PlonkProveWithPK_ms"], q["zk_bn254_plonk_prove_ms"])
PY
done
