#!/bin/bash
# round 3, final validation B: rocprofv3 kernel stats + traces (timelines), PMC traffic at 2^20 and 2^24
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3hb; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-host-inputs --no-micro > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
python3 $R/tools/summarize_rocprof.py $O/stats "bench.py --steps 50 --no-cpu-baseline --no-host-inputs --no-micro under rocprofv3 (Groth16 2^20 x 55, 2^24 x 6 + its two-slice check, PLONK 2^22 x 4)" > $O/bench_kernel_stats.md 2>&1
python3 $R/tools/timeline_proof.py $O/stats groth16 100 > $O/timeline_groth16_2p20.txt 2>&1
python3 $R/tools/timeline_proof.py $O/stats groth16_2p24 1000 > $O/timeline_groth16_2p24.txt 2>&1
python3 $R/tools/timeline_proof.py $O/stats plonk 300 > $O/timeline_plonk_2p22.txt 2>&1
rm -rf $O/stats
for ln in 20 24; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f$ln -o f -- python3 $R/bench.py --log-n $ln --steps 2 --warmup 1 --no-cpu-baseline --no-host-inputs --no-2p24 --no-plonk --no-micro > $O/pmc_f$ln.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w$ln -o w -- python3 $R/bench.py --log-n $ln --steps 2 --warmup 1 --no-cpu-baseline --no-host-inputs --no-2p24 --no-plonk --no-micro > $O/pmc_w$ln.log 2>&1
  cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null
  python3 $R/tools/pmc_traffic.py $O/pmc_f$ln $O/pmc_w$ln $R/profiles/pmc_traffic.json $ln "round 3 HEAD" > $O/pmc_traffic_$ln.txt 2>&1
  cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
  rm -rf $O/pmc_f$ln $O/pmc_w$ln
done
ls -la $O; cat $O/pmc_traffic_20.txt $O/pmc_traffic_24.txt
