#!/bin/bash
# Round 5, batch G: the compaction POLICY (a key remembers whether its last wire vector had zero digits): modes 0 never / 1 policy / 2 always on uniform and
# witness-like 2^20 proofs; the pipelined key read + session streams warmed beside it in the cold ProveWithPK; tests of everything touched.
set -u
O=gpurun_out/${1:-rnd5g}
mkdir -p $O /tmp/g16
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_keyio.py tests/test_gpu_goffi.py -m gpu -x -q > $O/pytest_keyio_goffi.txt 2>&1; echo "rc=$?" >> $O/pytest_keyio_goffi.txt; tail -4 $O/pytest_keyio_goffi.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "groth16_prove_vs_oracle or msm_witness or giant or golden_proofs or sharded_prove or rank_local or compact_key or 2p20_proof" > $O/pytest_sel.txt 2>&1; echo "rc=$?" >> $O/pytest_sel.txt; tail -4 $O/pytest_sel.txt
python tools/export_bench_groth16.py make /tmp/g16 20 > $O/g16_make.json 2> $O/g16_make.err
timeout 900 python tools/export_bench_groth16.py preprocess /tmp/g16 > $O/g16_preprocess.json 2> $O/g16_preprocess.err; echo "preprocess rc=$?"; tail -c 1500 $O/g16_preprocess.json; tail -3 $O/g16_preprocess.err
for rep in 1 2 3; do
timeout 900 python tools/export_bench_groth16.py prove /tmp/g16 10 > $O/g16_prove_$rep.json 2> $O/g16_prove.err; echo "prove rc=$?"; python - <<PY
import json
d=json.load(open("$O/g16_prove_$rep.json"))
print(d["cold_ProveWithPK_ms"], {k:v for k,v in d["cold_phases"].items() if v>3}, "warm", d["warm_ProveWithPK_ms"], d["warm_phases_per_call"].get("groth16_prove"))
PY
done
make -C noir_backend_using_gnark_amd/csrc EXPERIMENTS=1 -j16 > $O/make_exp.log 2>&1; echo "make exp rc=$?"
for rep in 1 2 3; do for mode in 1 0 2; do for sc in uniform witness; do
  ZKMI_W_DROP_ZERO_DIGITS=$mode timeout 600 python bench.py --lib exp --steps 100 --scalars $sc --no-2p24 --no-plonk --no-micro --no-export --no-cpu-baseline --no-host-inputs > $O/b.json 2> $O/b.err
  python -c "
import json;b=json.load(open('$O/b.json'));print(json.dumps({'mode':$mode,'scalars':'$sc','ms':b['ms_per_step'],'digits_ms':b['roofline']['kernel_ms_per_step'].get('msm_digits')}))" | tee -a $O/drop_policy_modes.jsonl
done; done; done
