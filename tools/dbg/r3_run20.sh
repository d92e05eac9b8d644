#!/bin/bash
# computeH with a and b in the same launches (ZKMI_H_BATCH=1): parity subset on the experiments build, then alternating A/B pairs
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3j; mkdir -p $O
cd $R
ZKMI_USE_EXPERIMENTS_LIB=1 ZKMI_H_BATCH=1 timeout 900 python -m pytest tests -m gpu -x -q -k "compute_h or golden or vs_oracle or groth16" > $O/pytest_hbatch.txt 2>&1; tail -3 $O/pytest_hbatch.txt
for i in 1 2 3; do
python tools/ab_bench.py $O/ab.jsonl --steps 100 --only 2p20 "base$i" "hbatch$i:ZKMI_H_BATCH=1" > /dev/null 2>&1
done
python tools/ab_bench.py $O/ab.jsonl --steps 6 --only 2p24 "base24" "hbatch24:ZKMI_H_BATCH=1" "base24b" "hbatch24b:ZKMI_H_BATCH=1" > /dev/null 2>&1
cut -c1-400 $O/ab.jsonl
