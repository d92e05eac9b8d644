#!/bin/bash
# computeH under the G2.B accumulate (after prepare(w) alone), with that kernel at full / half occupancy
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3p; mkdir -p $O
cd $R
for i in 1 2; do
python tools/ab_bench.py $O/ab.jsonl --steps 100 --only 2p20 "base$i" "under$i:ZKMI_H_UNDER_G2=1" "under_wg1_$i:ZKMI_H_UNDER_G2=1,ZKMI_ACC_WG_G2=1" "wg1_$i:ZKMI_ACC_WG_G2=1" > /dev/null 2>&1
done
python tools/ab_bench.py $O/ab.jsonl --steps 6 --only 2p24 "base24" "under24:ZKMI_H_UNDER_G2=1" "under_wg1_24:ZKMI_H_UNDER_G2=1,ZKMI_ACC_WG_G2=1" > /dev/null 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r3p/ab.jsonl'):
    d=json.loads(l); k=d.get('kernels_2p20',{})
    print(d['name'], d.get('prove_2p20_ms'), d.get('prove_2p24_ms'), d.get('parity_error'), {x:v for x,v in k.items() if 'ntt' in x or 'accumulate' in x or 'sort' in x})
PY
