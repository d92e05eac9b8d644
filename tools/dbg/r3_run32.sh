#!/bin/bash
# the task-planning chain with its single-workgroup steps run by the last workgroup of the launch before (ZKMI_PREP_MERGE, default on)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3w; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm or giant or skew or witness or collisions" > $O/pytest_product.txt 2>&1; tail -2 $O/pytest_product.txt
export ZKMI_BENCH_PLONK_REPS=8
for i in 1 2; do
python tools/ab_bench.py $O/ab.jsonl --steps 100 --only 2p20 --only plonk "chain$i:ZKMI_PREP_MERGE=0" "merged$i" > /dev/null 2>&1
done
python tools/ab_bench.py $O/ab.jsonl --steps 6 --only 2p24 --only micro "chain_b:ZKMI_PREP_MERGE=0" "merged_b" > /dev/null 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r3w/ab.jsonl'):
    d=json.loads(l); print(d['name'], {x:v for x,v in d.items() if not isinstance(v,(dict,list)) and x not in ('name','env','proof_sha','valu_frac')})
PY
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o st -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-inputs --no-2p24 --no-plonk --no-micro > $O/b.json 2> $O/b.err
python3 $R/tools/timeline_proof.py $O/tr groth16 3 > $O/timeline_groth16_2p20_fine.txt 2>&1
rm -rf $O/tr
awk '$1<1.5' $O/timeline_groth16_2p20_fine.txt | grep -v "^#" | tail -14 | cut -c1-90
