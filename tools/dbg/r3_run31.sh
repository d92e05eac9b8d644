#!/bin/bash
# PLONK with the transforms of the start of the day (experiments library linked with ntt.hip of 0bd549e) against HEAD's, alternating on one box, 12 proofs per figure
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3u; mkdir -p $O
cd $R
export ZKMI_BENCH_PLONK_REPS=12
for i in 1 2 3 4; do
python tools/ab_bench.py $O/ab2.jsonl --steps 20 --only plonk "old_ntt$i" "head$i:LIB=product" > /dev/null 2>&1
done
python3 - <<'PY'
import json
for l in open('gpurun_out/r3u/ab2.jsonl'):
    d=json.loads(l); pk=d.get('plonk_kernels',{}); print(d['name'], d.get('plonk_ms'), {k:v for k,v in pk.items() if 'ntt' in k or 'quot' in k or 'accum' in k})
PY
