#!/bin/bash
# round 3, final validation C: per-workload rocprofv3 kernel stats (the headline's own command, the 2^24 block, the PLONK block), stress runs
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3lc; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s20 -o st -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-host-inputs --no-2p24 --no-plonk --no-micro > $O/bench_2p20_under_rocprof.json 2> $O/err20.log
python3 $R/tools/summarize_rocprof.py $O/s20 "bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-host-inputs --no-2p24 --no-plonk --no-micro under rocprofv3: Groth16 prove at 2^20 constraints only (55 proofs + key load)" > $O/bench_2p20_kernel_stats.md 2>&1
rm -rf $O/s20
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s24 -o st -- python3 $R/bench.py --log-n 24 --steps 5 --warmup 1 --no-cpu-baseline --no-host-inputs --no-2p24 --no-plonk --no-micro > $O/bench_2p24_under_rocprof.json 2> $O/err24.log
python3 $R/tools/summarize_rocprof.py $O/s24 "bench.py --log-n 24 --steps 5 --warmup 1 (no other block) under rocprofv3: Groth16 prove at 2^24 constraints (6 proofs + key load with 84 GB of window tables)" > $O/bench_2p24_kernel_stats.md 2>&1
rm -rf $O/s24
cd $R
timeout 900 python tools/stress_prove.py 20 4000 > $O/stress_2p20.txt 2>&1; timeout 600 python tools/stress_plonk.py >> $O/stress_2p20.txt 2>&1; tail -4 $O/stress_2p20.txt
ls -la $O
