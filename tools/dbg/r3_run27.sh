#!/bin/bash
# the stage on index bit 0 without its product by the unit twiddle (default on; ZKMI_NTT_UNIT=0 in the experiments build restores it)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3q; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "ntt or compute_h or golden or prove_vs_oracle or groth16_2p20 or plonk_golden or sharded" > $O/pytest_product.txt 2>&1; tail -2 $O/pytest_product.txt
export ZKMI_USE_EXPERIMENTS_LIB=1
for rep in 1 2; do
ZKMI_NTT_UNIT=0 python tools/compute_h_bench.py >> $O/h.jsonl 2>$O/err.txt
python tools/compute_h_bench.py >> $O/h.jsonl 2>>$O/err.txt
done
python3 - <<'PY'
import json
for l in open('gpurun_out/r3q/h.jsonl'):
    d=json.loads(l); print({k:v for k,v in d['switches'].items() if k!='ZKMI_USE_EXPERIMENTS_LIB'}, [(k, d[k]['best_ms'], d[k]['median_ms'], d[k]['h_sha'][:6]) for k in ('2p20','2p22','2p24')])
PY
for i in 1 2; do
python tools/ab_bench.py $O/ab.jsonl --steps 100 --only 2p20 --only plonk --only micro "nounit$i:ZKMI_NTT_UNIT=0" "unit$i" > /dev/null 2>&1
done
python3 - <<'PY'
import json
for l in open('gpurun_out/r3q/ab.jsonl'):
    d=json.loads(l); print(d['name'], {k:v for k,v in d.items() if not isinstance(v,(dict,list)) and k not in ('name','env')})
PY
