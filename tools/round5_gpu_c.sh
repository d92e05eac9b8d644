#!/bin/bash
# Round 5, batch C: zero-digit compaction with the pair count left on the device (every test that runs a multi-exp), the Groth16 export worker at 2^20 again
# (gate-aware cut guesses, circuit upload beside the key's decoding), witness-like / uniform 2^20 proofs with and without the early enqueue of prepare(h).
set -u
O=gpurun_out/${1:-rnd5c}
mkdir -p $O /tmp/g16
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plonk.py tests/test_gpu_keyio.py tests/test_gpu_goffi.py tests/test_gpu_multidev.py -m gpu -x -q --durations=12 > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -22 $O/pytest.txt
python tools/export_bench_groth16.py make /tmp/g16 20 > $O/g16_make.json 2> $O/g16_make.err
timeout 900 python tools/export_bench_groth16.py preprocess /tmp/g16 > $O/g16_preprocess.json 2> $O/g16_preprocess.err; echo "preprocess rc=$?"; tail -c 1800 $O/g16_preprocess.json; tail -3 $O/g16_preprocess.err
timeout 900 python tools/export_bench_groth16.py prove /tmp/g16 10 > $O/g16_prove.json 2> $O/g16_prove.err; echo "prove rc=$?"; tail -c 3500 $O/g16_prove.json; tail -3 $O/g16_prove.err
timeout 300 python tools/export_bench_groth16.py verify /tmp/g16 > $O/g16_verify.json 2> $O/g16_verify.err; echo "verify rc=$?"; cat $O/g16_verify.json
g++ -O2 -std=c++17 tools/raw_lower_bench.cpp -lpthread -o /tmp/raw_lower_bench && /tmp/raw_lower_bench /tmp/g16/raw.json | tee $O/raw_lower_bench.json
make -C noir_backend_using_gnark_amd/csrc EXPERIMENTS=1 -j16 > $O/make_exp.log 2>&1; echo "make exp rc=$?"
for rep in 1 2; do for early in 0 1; do for sc in witness uniform; do
  ZKMI_PREPH_EARLY=$early timeout 600 python bench.py --lib exp --steps 60 --scalars $sc --no-2p24 --no-plonk --no-micro --no-export --no-cpu-baseline --no-host-inputs > $O/b.json 2> $O/b.err
  python -c "
import json;b=json.load(open('$O/b.json'));print(json.dumps({'preph_early':$early,'scalars':'$sc','ms':b['ms_per_step']}))" | tee -a $O/preph_early_ab.jsonl
done; done; done
for drop in 0 1; do for sc in witness uniform; do
  ZKMI_W_DROP_ZERO_DIGITS=$drop timeout 600 python bench.py --lib exp --steps 60 --scalars $sc --no-2p24 --no-plonk --no-micro --no-export --no-cpu-baseline --no-host-inputs > $O/b.json 2> $O/b.err
  python -c "
import json;b=json.load(open('$O/b.json'));print(json.dumps({'drop_zero_digits':$drop,'scalars':'$sc','ms':b['ms_per_step']}))" | tee -a $O/drop_zero_digits_ab.jsonl
done; done
