#!/bin/bash
# Round 5, batch X (experiment): a lean process with two streams of its own (slots 2-4 borrow them until the second proof) against five -- export tests first
set -u
O=gpurun_out/${1:-rnd5x}
mkdir -p $O /tmp/g16 /tmp/plk
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_goffi.py tests/test_gpu_keyio.py -m gpu -x -q > $O/pytest_sel.txt 2>&1; echo "rc=$?" >> $O/pytest_sel.txt; tail -3 $O/pytest_sel.txt
python tools/export_bench_groth16.py make /tmp/g16 20 > $O/g16_make.json 2> $O/g16_make.err
timeout 900 python tools/export_bench_groth16.py preprocess /tmp/g16 > $O/g16_preprocess.json 2> $O/g16_preprocess.err; echo "preprocess rc=$?"
python tools/export_bench.py make /tmp/plk > $O/plk_make.json 2> $O/plk_make.err
timeout 900 python tools/export_bench.py preprocess /tmp/plk > $O/plk_preprocess.json 2> $O/plk_preprocess.err; echo "plonk preprocess rc=$?"
for rep in 1 2 3 4; do
for n in 2 5 3; do
ZKMI_TMP_LEAN_STREAMS=$n timeout 900 python tools/export_bench_groth16.py prove /tmp/g16 6 > $O/g16_prove_${n}_$rep.json 2> $O/g16_prove.err; python - <<PY
import json
d=json.load(open("$O/g16_prove_${n}_$rep.json")); p=d["cold_phases"]; s=d["second_phases"]
print("streams $n: g16 cold", d["cold_ProveWithPK_ms"], "hip_init", p.get("hip_init"), "pk_read", p.get("pk_read"), "circuit", p.get("circuit_to_device"), "prove", p.get("groth16_prove"), "| second", d["second_ProveWithPK_ms"], "session_streams", s.get("session_streams"), "| warm", d["warm_ProveWithPK_ms"], "verifies", d["verifies"], d["warm_proof_verifies"])
PY
ZKMI_TMP_LEAN_STREAMS=$n timeout 900 python tools/export_bench.py prove /tmp/plk 4 > $O/plk_prove_${n}_$rep.json 2> $O/plk_prove.err; python - <<PY
import json
d=json.load(open("$O/plk_prove_${n}_$rep.json")); p=d["cold_phases"]; s=d["second_phases"]
print("streams $n: plonk cold", d["cold_PlonkProveWithPK_ms"], "hip_init", p.get("hip_init"), "srs_decode", p.get("srs_decode"), "pk_resident", p.get("pk_resident"), "prove", p.get("plonk_prove"), "| second", d["second_PlonkProveWithPK_ms"], "session_streams", s.get("session_streams"), "| warm", d["warm_PlonkProveWithPK_ms"], "verifies", d["verifies"])
PY
done
done
# the export blocks inside bench.py without / with the CPU baseline before them (the PLONK worker's warm call measured 19.3 ms inside the default line against 15.5 alone)
for flag in "--no-cpu-baseline" ""; do
timeout 1200 python bench.py --steps 20 --no-2p24 --no-plonk --no-micro --no-host-inputs $flag > $O/bench_small${flag}.json 2> $O/bench_small.err; python - <<PY
import json
d=json.loads([l for l in open("$O/bench_small${flag}.json") if l.startswith("{")][-1])
e=d.get("export_path",{}); p=e.get("prove_process",{})
print("bench $flag: plonk export warm", e.get("warm_PlonkProveWithPK_ms"), e.get("warm_over_prove"), p.get("warm_phases_per_call"))
e=d.get("export_path_groth16",{}); p=e.get("prove_process",{})
print("bench $flag: g16 export warm", e.get("warm_ProveWithPK_ms"), e.get("warm_over_prove"), p.get("warm_phases_per_call"))
PY
done
