#!/bin/bash
# Round 5, batch F: LDS-staged zero-digit compaction (A/B pairs), G2 decompression with the x0-based subgroup test and the complex-method square root,
# cooperative long rows in Setup's transposed products, lazy delta tables: the key / export tests and the 2^20 export worker again.
set -u
O=gpurun_out/${1:-rnd5f}
mkdir -p $O /tmp/g16
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_keyio.py tests/test_gpu_goffi.py -m gpu -x -q --durations=6 > $O/pytest_keyio_goffi.txt 2>&1; echo "rc=$?" >> $O/pytest_keyio_goffi.txt; tail -14 $O/pytest_keyio_goffi.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "groth16_prove_vs_oracle or msm_witness or batched_multi or giant or setup or r1cs or from_raw or golden_proofs or window_bits" > $O/pytest_sel.txt 2>&1; echo "rc=$?" >> $O/pytest_sel.txt; tail -4 $O/pytest_sel.txt
python tools/export_bench_groth16.py make /tmp/g16 20 > $O/g16_make.json 2> $O/g16_make.err
timeout 900 python tools/export_bench_groth16.py preprocess /tmp/g16 > $O/g16_preprocess.json 2> $O/g16_preprocess.err; echo "preprocess rc=$?"; tail -c 2600 $O/g16_preprocess.json; tail -3 $O/g16_preprocess.err
timeout 900 python tools/export_bench_groth16.py prove /tmp/g16 10 > $O/g16_prove.json 2> $O/g16_prove.err; echo "prove rc=$?"; tail -c 4200 $O/g16_prove.json; tail -3 $O/g16_prove.err
timeout 300 python tools/export_bench_groth16.py verify /tmp/g16 > $O/g16_verify.json 2> $O/g16_verify.err; echo "verify rc=$?"; cat $O/g16_verify.json
make -C noir_backend_using_gnark_amd/csrc EXPERIMENTS=1 -j16 > $O/make_exp.log 2>&1; echo "make exp rc=$?"
python tools/ab_bench.py $O/drop_zero_digits_pairs_lds_staged.jsonl --steps 100 --only 2p20 "drop1:ZKMI_W_DROP_ZERO_DIGITS=1" "drop0:ZKMI_W_DROP_ZERO_DIGITS=0" "drop1:ZKMI_W_DROP_ZERO_DIGITS=1" "drop0:ZKMI_W_DROP_ZERO_DIGITS=0" "drop1:ZKMI_W_DROP_ZERO_DIGITS=1" "drop0:ZKMI_W_DROP_ZERO_DIGITS=0" 2>&1 | cut -c1-300
for rep in 1 2 3; do for drop in 1 0; do
  ZKMI_W_DROP_ZERO_DIGITS=$drop timeout 600 python bench.py --lib exp --steps 100 --scalars witness --no-2p24 --no-plonk --no-micro --no-export --no-cpu-baseline --no-host-inputs > $O/b.json 2> $O/b.err
  python -c "
import json;b=json.load(open('$O/b.json'));print(json.dumps({'drop_zero_digits':$drop,'scalars':'witness','ms':b['ms_per_step'],'digits_ms':b['roofline']['kernel_ms_per_step'].get('msm_digits')}))" | tee -a $O/drop_zero_digits_witness.jsonl
done; done
