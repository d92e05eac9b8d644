#!/bin/bash
# G1 accumulate: the second point of a task through the shorter formula (product build) against the generic formula (experiments build made with -DZKMI_NO_SECOND)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3s; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm or groth16 or prove or collisions or giant or skew or witness or registered or opposite or window" > $O/pytest_product.txt 2>&1; tail -2 $O/pytest_product.txt
for i in 1 2 3; do
python tools/ab_bench.py $O/ab.jsonl --steps 100 --only 2p20 "generic$i" "second$i:LIB=product" > /dev/null 2>&1
done
python tools/ab_bench.py $O/ab.jsonl --steps 6 --only 2p24 --only plonk --only micro "generic_b" "second_b:LIB=product" "generic_c" "second_c:LIB=product" > /dev/null 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r3s/ab.jsonl'):
    d=json.loads(l); k=d.get('kernels_2p20',{})
    print(d['name'], {x:v for x,v in d.items() if not isinstance(v,(dict,list)) and x not in ('name','env','proof_sha')}, 'acc_g1', k.get('msm_accumulate_g1'))
PY
