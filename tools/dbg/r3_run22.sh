#!/bin/bash
# stand-alone computeH with the two experiment switches of ntt.hip
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3l; mkdir -p $O
cd $R
export ZKMI_USE_EXPERIMENTS_LIB=1
for rep in 1 2; do
python tools/compute_h_bench.py >> $O/h.jsonl 2>$O/err.txt
ZKMI_H_BATCH=1 python tools/compute_h_bench.py >> $O/h.jsonl 2>>$O/err.txt
ZKMI_H_FUSE_PW=1 python tools/compute_h_bench.py >> $O/h.jsonl 2>>$O/err.txt
ZKMI_H_BATCH=1 ZKMI_H_FUSE_PW=1 python tools/compute_h_bench.py >> $O/h.jsonl 2>>$O/err.txt
done
cat $O/h.jsonl; tail -3 $O/err.txt
