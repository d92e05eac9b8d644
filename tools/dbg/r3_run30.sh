#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3t; mkdir -p $O
cd $R
ZKMI_BENCH_DEBUG=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-2p24 --no-plonk --no-micro > $O/b.json 2> $O/b.err; grep "inner boundary" $O/b.err
