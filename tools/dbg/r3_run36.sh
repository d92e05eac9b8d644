#!/bin/bash
# accumulate kernels with __launch_bounds__(256) (experiments library built with -DZKMI_ACC_PLAIN_BOUNDS) against (256, 4 / 2) (product), alternating on one box
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3aa; mkdir -p $O
cd $R
for i in 1 2 3 4; do
python tools/ab_bench.py $O/ab.jsonl --steps 100 --only 2p20 "plain$i" "explicit$i:LIB=product" > /dev/null 2>&1
done
python tools/ab_bench.py $O/ab.jsonl --steps 6 --only 2p24 --only micro "plain_b" "explicit_b:LIB=product" "plain_c" "explicit_c:LIB=product" > /dev/null 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r3aa/ab.jsonl'):
    d=json.loads(l); k=d.get('kernels_2p20',{})
    print(d['name'], {x:v for x,v in d.items() if not isinstance(v,(dict,list)) and x not in ('name','env','proof_sha','parity_error')}, 'acc', k.get('msm_accumulate_g1'), k.get('msm_accumulate_g2'))
PY
