#!/bin/bash
# the 2^26 transform and computeH at 2^20..2^24: ntt.hip of the start of the day (linked into the experiments library) against HEAD's, alternating on one box
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3y; mkdir -p $O
cd $R
for i in 1 2 3; do
python tools/ab_bench.py $O/ab.jsonl --steps 10 --only micro "old_ntt$i" "head$i:LIB=product" > /dev/null 2>&1
ZKMI_USE_EXPERIMENTS_LIB=1 python tools/compute_h_bench.py >> $O/h.jsonl 2>/dev/null
python tools/compute_h_bench.py >> $O/h.jsonl 2>/dev/null
done
python3 - <<'PY'
import json
for l in open('gpurun_out/r3y/ab.jsonl'):
    d=json.loads(l); print(d['name'], d.get('ntt_2p26_ms'), d.get('msm_2p26_ms'))
for l in open('gpurun_out/r3y/h.jsonl'):
    d=json.loads(l); print('old' if d['switches'].get('ZKMI_USE_EXPERIMENTS_LIB') else 'head', [(k, d[k]['best_ms'], d[k]['median_ms']) for k in ('2p20','2p22','2p24')])
PY
