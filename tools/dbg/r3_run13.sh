#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3m; mkdir -p $O
cd $R
ZKMI_USE_EXPERIMENTS_LIB=1 python -m pytest tests/test_gpu_parity.py -q -m gpu -x \
  -k "msm_golden or vs_oracle_uniform or witness_like or giant or collisions or groth16_golden or prove_vs_oracle or registered_bases or equal_and_opposite or window_bits" > $O/t_pair.log 2>&1; tail -3 $O/t_pair.log
python tools/ab_bench.py $O/ab.jsonl --steps 60 "product:LIB=product" "pair" "product2:LIB=product" "pair2" 2>&1 | tail -5
