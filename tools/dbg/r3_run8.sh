#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3h; mkdir -p $O
cd $R
python tools/ab_bench.py $O/ab.jsonl --steps 60 --only plonk "base" "tail_exposed:ZKMI_FUSE_TAIL=2" "base2" "tail_exposed2:ZKMI_FUSE_TAIL=2" "tail_exposed_fold:ZKMI_FUSE_TAIL=2,ZKMI_FUSE_FOLD=1" 2>&1 | tail -6
bash tools/dbg/r3_final_b.sh
