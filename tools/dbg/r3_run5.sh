#!/bin/bash
# round 3, GPU call 5: product library without rocPRIM (parity), schedule switches with priorities, full default bench line
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3e; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py -q -m gpu -x \
  -k "msm_golden or vs_oracle_uniform or witness_like or giant or g2_msm_vs or collisions or groth16_golden or prove_vs_oracle or registered_bases or equal_and_opposite or window_bits or compact_key or 2p22" > $O/t_product.log 2>&1; tail -3 $O/t_product.log
python tools/ab_bench.py $O/ab.jsonl --steps 40 "product:LIB=product" "base" "nogate:ZKMI_NOGATE=1" "chain:ZKMI_CHAIN=1" "nogate_chain:ZKMI_NOGATE=1,ZKMI_CHAIN=1" "preph_low:ZKMI_PREPH_LOW=1" 2>&1 | tail -8
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
