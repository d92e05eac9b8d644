#!/bin/bash
# round 3, GPU call 3: the hand-written sort (stand-alone check + timing, parity inside the MSM, A/B), kernel timelines (csv), FETCH_SIZE calibration
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3c; mkdir -p $O
cd $R
timeout 600 tools/rs_test > $O/rs_test.log 2>&1; tail -12 $O/rs_test.log
ZKMI_USE_EXPERIMENTS_LIB=1 ZKMI_SORT=1 python -m pytest tests/test_gpu_parity.py -q -m gpu -x \
  -k "msm_golden or vs_oracle_uniform or witness_like or giant or g2_msm_vs or collisions or groth16_golden or prove_vs_oracle or registered_bases or equal_and_opposite or window_bits" > $O/t_sort.log 2>&1; tail -3 $O/t_sort.log
python tools/ab_bench.py $O/ab.jsonl --steps 40 "prio3" "prio3_sort:ZKMI_SORT=1" "prio3_notables_sort:ZKMI_SORT=1,ZKMI_TABLE_CAP_GB=0" 2>&1 | tail -8
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o tr -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-host-inputs --no-2p24 --no-micro > $O/trace_bench.log 2>&1
find $O/trace -name "*kernel_trace.csv" | head -1 | xargs -I{} sh -c 'gzip -c {} > '$O'/kernel_trace_product.csv.gz'
ZKMI_USE_EXPERIMENTS_LIB=1 ZKMI_SORT=1 rocprofv3 --kernel-trace --output-format csv -d $O/trace2 -o tr -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-host-inputs --no-2p24 --no-micro > $O/trace2_bench.log 2>&1
find $O/trace2 -name "*kernel_trace.csv" | head -1 | xargs -I{} sh -c 'gzip -c {} > '$O'/kernel_trace_prio3_sort.csv.gz'
rm -rf $O/trace $O/trace2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib -o c -- $R/tools/gather_calib > $O/calib.log 2>&1
python3 $R/tools/gather_calib_summary.py $O/calib > $O/gather_calib.json 2>&1; cat $O/gather_calib.json | head -40
rm -rf $O/calib
ls -la $O
