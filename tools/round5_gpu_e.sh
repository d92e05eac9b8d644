#!/bin/bash
# Round 5, batch E: single-pass zero-digit compaction (A/B in alternating pairs, uniform and witness-like), the MFMA-route decision benchmark (tools/ubench5.hip),
# the Groth16 export worker with per-kernel totals of the cold calls, the default bench line, PMC traffic of the accumulate kernels (2^20).
set -u
O=gpurun_out/${1:-rnd5e}
mkdir -p $O /tmp/g16
export TMPDIR=/tmp
( cd tools && hipcc -O3 --offload-arch=gfx950 -I../noir_backend_using_gnark_amd/csrc ubench5.hip -o /tmp/ubench5 && timeout 300 /tmp/ubench5 ) > $O/mfma_route_ubench.json 2> $O/ubench5.err; cat $O/mfma_route_ubench.json
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "groth16_prove_vs_oracle or msm_witness or batched_multi or giant or 2p20_proof" > $O/pytest_sel.txt 2>&1; echo "rc=$?" >> $O/pytest_sel.txt; tail -4 $O/pytest_sel.txt
timeout 600 python -m pytest tests/test_gpu_plonk.py -m gpu -x -q -k "random_circuits or lagrange or sparse or wire" > $O/pytest_plonk_sel.txt 2>&1; echo "rc=$?" >> $O/pytest_plonk_sel.txt; tail -4 $O/pytest_plonk_sel.txt
python tools/export_bench_groth16.py make /tmp/g16 20 > $O/g16_make.json 2> $O/g16_make.err
timeout 900 python tools/export_bench_groth16.py preprocess /tmp/g16 > $O/g16_preprocess.json 2> $O/g16_preprocess.err; echo "preprocess rc=$?"; tail -c 2600 $O/g16_preprocess.json; tail -3 $O/g16_preprocess.err
timeout 900 python tools/export_bench_groth16.py prove /tmp/g16 10 > $O/g16_prove.json 2> $O/g16_prove.err; echo "prove rc=$?"; tail -c 4200 $O/g16_prove.json; tail -3 $O/g16_prove.err
make -C noir_backend_using_gnark_amd/csrc EXPERIMENTS=1 -j16 > $O/make_exp.log 2>&1; echo "make exp rc=$?"
python tools/ab_bench.py $O/drop_zero_digits_pairs_single_pass.jsonl --steps 100 --only 2p20 "drop1:ZKMI_W_DROP_ZERO_DIGITS=1" "drop0:ZKMI_W_DROP_ZERO_DIGITS=0" "drop1:ZKMI_W_DROP_ZERO_DIGITS=1" "drop0:ZKMI_W_DROP_ZERO_DIGITS=0" "drop1:ZKMI_W_DROP_ZERO_DIGITS=1" "drop0:ZKMI_W_DROP_ZERO_DIGITS=0" 2>&1 | cut -c1-300
for rep in 1 2 3; do for drop in 1 0; do
  ZKMI_W_DROP_ZERO_DIGITS=$drop timeout 600 python bench.py --lib exp --steps 100 --scalars witness --no-2p24 --no-plonk --no-micro --no-export --no-cpu-baseline --no-host-inputs > $O/b.json 2> $O/b.err
  python -c "
import json;b=json.load(open('$O/b.json'));print(json.dumps({'drop_zero_digits':$drop,'scalars':'witness','ms':b['ms_per_step']}))" | tee -a $O/drop_zero_digits_witness.jsonl
done; done
timeout 1500 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<PY
import json
d=json.loads([l for l in open("$O/bench_default_line.json") if l.startswith("{")][-1])
print({k:d[k] for k in ("value","ms_per_step","vs_baseline")}, d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"].get("measured_in"))
for k in ("prove_ms_host_inputs","prove_ms_witness_like_scalars"): print(k, d.get(k,{}).get("value"))
print("2p24", d.get("at_2p24",{}).get("prove_ms"), "plonk", [d[k]["prove_ms"] for k in d if k.startswith("plonk_2p")])
e=d.get("export_path_groth16",{}); print("g16 export", {k:e.get(k) for k in ("cold_ProveWithPK_ms","warm_ProveWithPK_ms","zk_bn254_groth16_prove_r1cs_ms","warm_over_prove","ok","error")})
e=d.get("export_path",{}); print("plonk export", {k:e.get(k) for k in ("warm_PlonkProveWithPK_ms","warm_over_prove","ok","error")})
print("parity_error", d.get("parity_error"), "cpu", d.get("cpu_baseline",{}).get("prove_ms"))
PY
