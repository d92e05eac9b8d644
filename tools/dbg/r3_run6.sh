#!/bin/bash
# round 3, GPU call 6: prepare(h) placement at 2^24, 2^24 timeline
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3f; mkdir -p $O
cd $R
python tools/ab_bench.py $O/ab.jsonl --steps 30 --only 2p24 "base" "preph_first:ZKMI_PREPH_FIRST=1" "nogate:ZKMI_NOGATE=1" 2>&1 | tail -4
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o tr -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-inputs --no-plonk --no-micro > $O/trace_bench.log 2>&1
find $O/trace -name "*kernel_trace.csv" | head -1 | xargs -I{} sh -c 'gzip -c {} > '$O'/kernel_trace_2p24.csv.gz'
rm -rf $O/trace
ls -la $O
