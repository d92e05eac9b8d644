#!/bin/bash
# computeH: a, b in the same launches (ZKMI_H_BATCH=1) and the two element-wise steps inside the closing transform (ZKMI_H_FUSE_PW=1)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3k; mkdir -p $O
cd $R
ZKMI_USE_EXPERIMENTS_LIB=1 ZKMI_H_BATCH=1 ZKMI_H_FUSE_PW=1 timeout 900 python -m pytest tests -m gpu -x -q -k "compute_h or golden or prove_vs_oracle or groth16_2p20" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for i in 1 2 3; do
python tools/ab_bench.py $O/ab.jsonl --steps 100 --only 2p20 "base$i" "fuse$i:ZKMI_H_FUSE_PW=1" "both$i:ZKMI_H_BATCH=1,ZKMI_H_FUSE_PW=1" > /dev/null 2>&1
done
python tools/ab_bench.py $O/ab.jsonl --steps 6 --only 2p24 "base24" "fuse24:ZKMI_H_FUSE_PW=1" "both24:ZKMI_H_BATCH=1,ZKMI_H_FUSE_PW=1" "base24b" "fuse24b:ZKMI_H_FUSE_PW=1" "both24b:ZKMI_H_BATCH=1,ZKMI_H_FUSE_PW=1" > /dev/null 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r3k/ab.jsonl'):
    d=json.loads(l); k=d.get('kernels_2p20',{})
    print(d['name'], d.get('prove_2p20_ms'), d.get('prove_2p24_ms'), d.get('parity_error'), d.get('ok_2p24'), 'ntt', [k.get(x) for x in ('ntt_pass_strided','ntt_pass_contig_if','ntt_pass_contig','fr_mul','h_final')])
PY
