#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3n; mkdir -p $O
cd $R
ZKMI_USE_EXPERIMENTS_LIB=1 ZKMI_ACC_WG_G1=4 ZKMI_ACC_WG_G2=2 ZKMI_ACC_SERP=1 python -m pytest tests/test_gpu_parity.py -q -m gpu -x \
  -k "msm_golden or vs_oracle_uniform or witness_like or giant or g2_msm_vs or collisions or groth16_golden or prove_vs_oracle or registered_bases or equal_and_opposite or window_bits" > $O/t_serp.log 2>&1; tail -3 $O/t_serp.log
python tools/ab_bench.py $O/ab.jsonl --steps 60 "base" "serp4:ZKMI_ACC_WG_G1=4,ZKMI_ACC_SERP=1" "serp4_g2:ZKMI_ACC_WG_G1=4,ZKMI_ACC_WG_G2=2,ZKMI_ACC_SERP=1" "dyn4:ZKMI_ACC_WG_G1=4" "dyn4_g2:ZKMI_ACC_WG_G1=4,ZKMI_ACC_WG_G2=2" "base2" 2>&1 | tail -7
