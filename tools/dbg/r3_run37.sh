#!/bin/bash
# the radix sort with 256-thread workgroups and 4096-pair tiles (experiments library built with -DZKMI_RS_THREADS=256) against 512 / 8192 (product), alternating on one box
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3ab; mkdir -p $O
cd $R
ZKMI_USE_EXPERIMENTS_LIB=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm_golden or vs_oracle_uniform or witness_like or giant or collisions or skew" > $O/pytest.txt 2>&1; tail -1 $O/pytest.txt
export ZKMI_BENCH_PLONK_REPS=8
for i in 1 2 3; do
python tools/ab_bench.py $O/ab.jsonl --steps 100 --only 2p20 --only plonk "rs256_$i" "rs512_$i:LIB=product" > /dev/null 2>&1
done
python tools/ab_bench.py $O/ab.jsonl --steps 6 --only 2p24 --only micro "rs256_b" "rs512_b:LIB=product" "rs256_c" "rs512_c:LIB=product" > /dev/null 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r3ab/ab.jsonl'):
    d=json.loads(l); k=d.get('kernels_2p20',{})
    print(d['name'], {x:v for x,v in d.items() if not isinstance(v,(dict,list)) and x not in ('name','env','proof_sha','parity_error','valu_frac')}, 'sort', k.get('msm_sort_pass'), k.get('msm_sort_hist'))
PY
