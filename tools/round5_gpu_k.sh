#!/bin/bash
# Round 5, batch K: the full -m gpu suite at HEAD, witness-like 2^24 with and without the compaction, the Groth16 export worker once more (content keys on 32 threads).
set -u
O=gpurun_out/${1:-rnd5k}
mkdir -p $O /tmp/g16
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest_full.txt 2>&1; echo "rc=$?" >> $O/pytest_full.txt; tail -16 $O/pytest_full.txt
python tools/export_bench_groth16.py make /tmp/g16 20 > $O/g16_make.json 2> $O/g16_make.err
timeout 900 python tools/export_bench_groth16.py preprocess /tmp/g16 > $O/g16_preprocess.json 2> $O/g16_preprocess.err; echo "preprocess rc=$?"
timeout 900 python tools/export_bench_groth16.py prove /tmp/g16 20 > $O/g16_prove.json 2> $O/g16_prove.err; echo "prove rc=$?"; python - <<PY
import json
d=json.load(open("$O/g16_prove.json")); print("cold", d["cold_ProveWithPK_ms"], "warm", d["warm_ProveWithPK_ms"], d["warm_phases_per_call"])
PY
make -C noir_backend_using_gnark_amd/csrc EXPERIMENTS=1 -j16 > $O/make_exp.log 2>&1; echo "make exp rc=$?"
for rep in 1 2; do for drop in 1 0; do
  ZKMI_W_DROP_ZERO_DIGITS=$drop timeout 900 python bench.py --lib exp --steps 20 --scalars witness --no-plonk --no-micro --no-export --no-cpu-baseline --no-host-inputs > $O/b.json 2> $O/b.err
  python -c "
import json;b=json.load(open('$O/b.json'));print(json.dumps({'drop_zero_digits':$drop,'scalars':'witness','ms_2p20':b['ms_per_step'],'ms_2p24':b['at_2p24']['prove_ms']}))" | tee -a $O/drop_zero_digits_witness_2p24.jsonl
done; done
