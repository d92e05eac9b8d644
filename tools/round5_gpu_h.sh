#!/bin/bash
# Round 5, batch H: the unmerged / uninitialised RawR1CS reader and the three-thread start of the Groth16 exports (cold calls, three runs each), compaction always on.
set -u
O=gpurun_out/${1:-rnd5h}
mkdir -p $O /tmp/g16
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_keyio.py tests/test_gpu_goffi.py -m gpu -x -q > $O/pytest_keyio_goffi.txt 2>&1; echo "rc=$?" >> $O/pytest_keyio_goffi.txt; tail -4 $O/pytest_keyio_goffi.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "groth16_prove_vs_oracle or msm_witness or giant or prepared or from_raw or r1cs" > $O/pytest_sel.txt 2>&1; echo "rc=$?" >> $O/pytest_sel.txt; tail -4 $O/pytest_sel.txt
python tools/export_bench_groth16.py make /tmp/g16 20 > $O/g16_make.json 2> $O/g16_make.err
g++ -O2 -std=c++17 tools/raw_lower_bench.cpp -lpthread -o /tmp/raw_lower_bench && /tmp/raw_lower_bench /tmp/g16/raw.json | tee $O/raw_lower_bench.json
for rep in 1 2 3; do
timeout 900 python tools/export_bench_groth16.py preprocess /tmp/g16 > $O/g16_preprocess_$rep.json 2> $O/g16_preprocess.err; echo "preprocess rc=$?"
timeout 900 python tools/export_bench_groth16.py prove /tmp/g16 10 > $O/g16_prove_$rep.json 2> $O/g16_prove.err; echo "prove rc=$?"
timeout 300 python tools/export_bench_groth16.py verify /tmp/g16 > $O/g16_verify_$rep.json 2> $O/g16_verify.err; echo "verify rc=$?"
python - <<PY
import json
p=json.load(open("$O/g16_preprocess_$rep.json")); d=json.load(open("$O/g16_prove_$rep.json")); v=json.load(open("$O/g16_verify_$rep.json"))
print("Preprocess", p["Preprocess_ms"], {k:x for k,x in p["phases"].items() if x>8})
print("cold prove", d["cold_ProveWithPK_ms"], {k:x for k,x in d["cold_phases"].items() if x>3}, "second", d["second_ProveWithPK_ms"], "warm", d["warm_ProveWithPK_ms"], d["warm_phases_per_call"])
print("cold verify", v["cold_VerifyWithVK_ms"], v["cold_phases"], "warm", v["second_VerifyWithVK_ms"])
PY
done
