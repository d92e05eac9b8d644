#!/bin/bash
# computeH: a, b, c in shared launches (ZKMI_H_BATCH=1, c in the inverse half) + fused element-wise steps (ZKMI_H_FUSE_PW=1)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3m; mkdir -p $O
cd $R
export ZKMI_USE_EXPERIMENTS_LIB=1
ZKMI_H_BATCH=1 ZKMI_H_FUSE_PW=1 timeout 900 python -m pytest tests -m gpu -x -q -k "compute_h or golden or prove_vs_oracle or groth16_2p20" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for rep in 1 2; do
python tools/compute_h_bench.py >> $O/h.jsonl 2>$O/err.txt
ZKMI_H_BATCH=1 python tools/compute_h_bench.py >> $O/h.jsonl 2>>$O/err.txt
ZKMI_H_BATCH=1 ZKMI_H_FUSE_PW=1 python tools/compute_h_bench.py >> $O/h.jsonl 2>>$O/err.txt
done
cat $O/h.jsonl; tail -3 $O/err.txt
for i in 1 2 3; do
python tools/ab_bench.py $O/ab.jsonl --steps 100 --only 2p20 "base$i" "both$i:ZKMI_H_BATCH=1,ZKMI_H_FUSE_PW=1" > /dev/null 2>&1
done
python3 - <<'PY'
import json
for l in open('gpurun_out/r3m/ab.jsonl'):
    d=json.loads(l); print(d['name'], d.get('prove_2p20_ms'), d.get('parity_error'))
PY
