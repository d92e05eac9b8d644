#!/bin/bash
# round 3, GPU call 2: new tests, parity of the fused tail / fold kernels, priority + fusion A/B, kernel timelines, FETCH_SIZE calibration
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3b; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_plonk.py -q -m gpu -k "handle_values" > $O/t_plonk.log 2>&1; tail -3 $O/t_plonk.log
python -m pytest tests/test_gpu_goffi.py -q -m gpu > $O/t_goffi.log 2>&1; tail -3 $O/t_goffi.log
ZKMI_USE_EXPERIMENTS_LIB=1 ZKMI_FUSE_TAIL=1 ZKMI_FUSE_FOLD=1 python -m pytest tests/test_gpu_parity.py -q -m gpu -x \
  -k "msm_golden or vs_oracle_uniform or witness_like or giant or g2_msm_vs or collisions or groth16_golden or prove_vs_oracle or registered_bases or equal_and_opposite or window_bits" > $O/t_fused.log 2>&1; tail -3 $O/t_fused.log
python tools/ab_bench.py $O/ab.jsonl --steps 40 "product:LIB=product" "prio3" "prio3_fold:ZKMI_FUSE_FOLD=1" "prio3_tail:ZKMI_FUSE_TAIL=1" "prio3_both:ZKMI_FUSE_FOLD=1,ZKMI_FUSE_TAIL=1" 2>&1 | tail -8
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/trace -o tr -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-host-inputs --no-2p24 --no-micro > $O/trace_bench.log 2>&1
find $O/trace -name "*kernel_trace.csv" | head -1 | xargs -I{} sh -c 'gzip -c {} > '$O'/kernel_trace_product.csv.gz'
ZKMI_USE_EXPERIMENTS_LIB=1 rocprofv3 --kernel-trace -d $O/trace2 -o tr -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-host-inputs --no-2p24 --no-micro > $O/trace2_bench.log 2>&1
find $O/trace2 -name "*kernel_trace.csv" | head -1 | xargs -I{} sh -c 'gzip -c {} > '$O'/kernel_trace_prio3.csv.gz'
rm -rf $O/trace $O/trace2
rocprofv3 --pmc FETCH_SIZE -d $O/calib -o c -- $R/tools/gather_calib > $O/calib.log 2>&1
python3 $R/tools/gather_calib_summary.py $O/calib > $O/gather_calib.json 2>&1; cat $O/gather_calib.json | head -40
rm -rf $O/calib
ls -la $O
