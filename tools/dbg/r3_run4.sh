#!/bin/bash
# round 3, GPU call 4: accumulate kernels with LDS-DMA prefetch of the next point (parity + A/B)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3d; mkdir -p $O
cd $R
ZKMI_USE_EXPERIMENTS_LIB=1 ZKMI_ACC_PF_G1=1 ZKMI_ACC_PF_G2=1 python -m pytest tests/test_gpu_parity.py -q -m gpu -x \
  -k "msm_golden or vs_oracle_uniform or witness_like or giant or g2_msm_vs or collisions or groth16_golden or prove_vs_oracle or registered_bases or equal_and_opposite or window_bits or compact_key" > $O/t_pf.log 2>&1; tail -3 $O/t_pf.log
python tools/ab_bench.py $O/ab.jsonl --steps 40 "product:LIB=product" "base" "pf_g1:ZKMI_ACC_PF_G1=1" "pf_g2:ZKMI_ACC_PF_G2=1" "pf_both:ZKMI_ACC_PF_G1=1,ZKMI_ACC_PF_G2=1" 2>&1 | tail -8
