#!/bin/bash
# round 3, final validation D: the product library with the graded wave priorities -- parity subset, smoke, default bench line
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3fd; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_plonk.py -q -m gpu -x -k "msm_golden or vs_oracle_uniform or witness_like or giant or g2_msm_vs or collisions or groth16_golden or prove_vs_oracle or registered_bases or equal_and_opposite or window_bits or compact_key or ntt_vs_oracle or compute_h or reference_fixtures or handle_values or golden" > $O/gpu_pytest_subset.log 2>&1; tail -3 $O/gpu_pytest_subset.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
