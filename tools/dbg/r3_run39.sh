#!/bin/bash
# one NTT pass workgroup per CU (12 KB of unused LDS each) so that a sort workgroup fits beside it while computeH and prepare(w) share the machine
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3ae; mkdir -p $O
cd $R
for i in 1 2 3; do
python tools/ab_bench.py $O/ab.jsonl --steps 100 --only 2p20 "base$i" "pad12k_$i:ZKMI_NTT_LDS_PAD=12288" > /dev/null 2>&1
done
python3 - <<'PY'
import json
for l in open('gpurun_out/r3ae/ab.jsonl'):
    d=json.loads(l); k=d.get('kernels_2p20',{}); print(d['name'], d.get('prove_2p20_ms'), d.get('parity_error'), {x:v for x,v in k.items() if 'ntt' in x or 'sort' in x})
PY
