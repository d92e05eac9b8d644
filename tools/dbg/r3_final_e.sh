#!/bin/bash
# round 3, final validation E: stress runs on the last build (the PLONK quotient kernel changed after the previous ones)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3ke; mkdir -p $O
cd $R
timeout 900 python tools/stress_prove.py 20 2000 > $O/stress.txt 2>&1; timeout 900 python tools/stress_plonk.py >> $O/stress.txt 2>&1; tail -4 $O/stress.txt
