#!/bin/bash
# PLONK's quotient numerator with its products in the 29-bit-limb representation (ZKMI_PLONK_QUOT29, default on) against the saturated form
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3ad; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_plonk.py tests/test_gpu_goffi.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
export ZKMI_BENCH_PLONK_REPS=10
for i in 1 2 3; do
python tools/ab_bench.py $O/ab.jsonl --steps 20 --only plonk "sat$i:ZKMI_PLONK_QUOT29=0" "u29_$i" > /dev/null 2>&1
done
python3 - <<'PY'
import json
for l in open('gpurun_out/r3ad/ab.jsonl'):
    d=json.loads(l); pk=d.get('plonk_kernels',{}); print(d['name'], d.get('plonk_ms'), d.get('plonk_ok'), 'quotient', pk.get('plonk_quotient'))
PY
