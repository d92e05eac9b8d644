#!/bin/bash
# round 3, final validation B2: the kernel timelines of one proof each (the anchor of tools/timeline_proof.py follows the fused closing transform)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3lb; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/stats -o st -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-host-inputs --no-micro > $O/bench_under_rocprof_b2.json 2> $O/bench_under_rocprof_b2.err
head -3 $(ls $O/stats/*kernel_trace.csv $O/stats/*/*kernel_trace.csv 2>/dev/null | head -1) | cut -c1-400
grep -m2 "k_ntt_pass29<" $(ls $O/stats/*kernel_trace.csv $O/stats/*/*kernel_trace.csv 2>/dev/null | head -1) | cut -c1-400
python3 $R/tools/timeline_proof.py $O/stats groth16 100 > $O/timeline_groth16_2p20.txt 2>&1; python3 $R/tools/timeline_proof.py $O/stats groth16 3 > $O/timeline_groth16_2p20_fine.txt 2>&1
python3 $R/tools/timeline_proof.py $O/stats groth16_2p24 1000 > $O/timeline_groth16_2p24.txt 2>&1
python3 $R/tools/timeline_proof.py $O/stats plonk 300 > $O/timeline_plonk_2p22.txt 2>&1
rm -rf $O/stats
wc -l $O/timeline_*.txt
