#!/bin/bash
# k_quotient29 with __launch_bounds__(256) (130 VGPRs, three waves per SIMD, no spills: experiments library) against (256, 4) (128 VGPRs, five spilled: product)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3ag; mkdir -p $O
cd $R
export ZKMI_BENCH_PLONK_REPS=8
for i in 1 2 3; do
python tools/ab_bench.py $O/ab.jsonl --steps 20 --only plonk "plain$i" "four_waves$i:LIB=product" > /dev/null 2>&1
done
python3 - <<'PY'
import json
for l in open('gpurun_out/r3ag/ab.jsonl'):
    d=json.loads(l); pk=d.get('plonk_kernels',{}); print(d['name'], d.get('plonk_ms'), 'quotient', pk.get('plonk_quotient'))
PY
