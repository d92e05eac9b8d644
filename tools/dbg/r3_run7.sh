#!/bin/bash
# round 3, GPU call 7: PLONK suite after the batched evaluations; resident accumulate grids vs the new sort at 2^24 / PLONK; FP64 product bound
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3g; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_plonk.py -q -m gpu -x -k "not 2p22" > $O/t_plonk.log 2>&1; tail -3 $O/t_plonk.log
tools/ubench4 > $O/ubench4.json 2>&1; cat $O/ubench4.json
python tools/ab_bench.py $O/ab.jsonl --steps 30 --only 2p24 --only plonk "base" "g1_3:ZKMI_ACC_WG_G1=3" "g1_3_g2_1:ZKMI_ACC_WG_G1=3,ZKMI_ACC_WG_G2=1" "g1_2_g2_1:ZKMI_ACC_WG_G1=2,ZKMI_ACC_WG_G2=1" 2>&1 | tail -5
