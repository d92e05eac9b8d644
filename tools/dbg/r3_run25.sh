#!/bin/bash
# computeH: a, b, c in shared launches (ZKMI_H_BATCH=1) + fused element-wise steps (ZKMI_H_FUSE_PW=1), after the scratch-copy fix
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3o; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -x -q -k "ntt or compute_h" > $O/pytest_product.txt 2>&1; tail -2 $O/pytest_product.txt
export ZKMI_USE_EXPERIMENTS_LIB=1
ZKMI_H_BATCH=1 ZKMI_H_FUSE_PW=1 timeout 900 python -m pytest tests -m gpu -x -q -k "ntt or compute_h or golden or prove_vs_oracle or groth16_2p20" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for rep in 1 2; do
python tools/compute_h_bench.py >> $O/h.jsonl 2>$O/err.txt
ZKMI_H_BATCH=1 python tools/compute_h_bench.py >> $O/h.jsonl 2>>$O/err.txt
ZKMI_H_FUSE_PW=1 python tools/compute_h_bench.py >> $O/h.jsonl 2>>$O/err.txt
ZKMI_H_BATCH=1 ZKMI_H_FUSE_PW=1 python tools/compute_h_bench.py >> $O/h.jsonl 2>>$O/err.txt
done
python3 - <<'PY'
import json
for l in open('gpurun_out/r3o/h.jsonl'):
    d=json.loads(l); print({k:v for k,v in d['switches'].items() if k!='ZKMI_USE_EXPERIMENTS_LIB'}, [(k, d[k]['best_ms'], d[k]['median_ms'], d[k]['h_sha'][:6]) for k in ('2p20','2p22','2p24')])
PY
for i in 1 2 3; do
python tools/ab_bench.py $O/ab.jsonl --steps 100 --only 2p20 "base$i" "batch$i:ZKMI_H_BATCH=1" "both$i:ZKMI_H_BATCH=1,ZKMI_H_FUSE_PW=1" > /dev/null 2>&1
done
python tools/ab_bench.py $O/ab.jsonl --steps 6 --only 2p24 --only plonk "base24" "both24:ZKMI_H_BATCH=1,ZKMI_H_FUSE_PW=1" "base24b" "both24b:ZKMI_H_BATCH=1,ZKMI_H_FUSE_PW=1" > /dev/null 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r3o/ab.jsonl'):
    d=json.loads(l); print(d['name'], d.get('prove_2p20_ms'), d.get('prove_2p24_ms'), d.get('plonk_2p22_ms'), d.get('parity_error'))
PY
