#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3l; mkdir -p $O
cd $R
python tools/ab_bench.py $O/ab.jsonl --steps 40 "base" "g1r2:ZKMI_ACC_WG_G1=2" "g1r2_pf:ZKMI_ACC_WG_G1=2,ZKMI_ACC_PF_G1=1" "g1r2_pf_chain:ZKMI_ACC_WG_G1=2,ZKMI_ACC_PF_G1=1,ZKMI_PLONK_CHAIN=1" "g1r3_pf:ZKMI_ACC_WG_G1=3,ZKMI_ACC_PF_G1=1" "all_r_pf:ZKMI_ACC_WG_G1=2,ZKMI_ACC_PF_G1=1,ZKMI_ACC_WG_G2=1,ZKMI_ACC_PF_G2=1" "g1r2_pf_g2pf:ZKMI_ACC_WG_G1=2,ZKMI_ACC_PF_G1=1,ZKMI_ACC_PF_G2=1" 2>&1 | tail -9
