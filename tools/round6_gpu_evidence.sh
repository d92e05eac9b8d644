#!/bin/bash
# Round 6, evidence at HEAD: the full -m gpu suite, smoke, the default bench line, the self-launched --gpus 2 line, rocprofv3 kernel stats of the 2^20 block and of the
# PLONK block, PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes, nothing but --pmc) at 2^20, 2^24 and for PLONK alone.
set -u
O=gpurun_out/${1:-r6ev}
mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --durations=15 ) > $O/pytest_full.txt 2>&1; echo "rc=$?" >> $O/pytest_full.txt; tail -24 $O/pytest_full.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
( time timeout 1500 python3 bench.py --steps 20 --warmup 5 ) > $O/bench_default_line.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -3 $O/bench_default.err
timeout 900 python3 bench.py --gpus 2 --log-n 16 --steps 3 --warmup 1 > $O/bench_gpus2_self_launch.json 2> $O/bench_gpus2.err; echo "bench --gpus 2 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/$O/prof_2p20 -- python3 bench.py --steps 100 --no-2p24 --no-plonk --no-micro --no-export --no-cpu-baseline --no-host-inputs > $O/bench_2p20_under_rocprof.json 2> $O/rocprof_2p20.err
python tools/summarize_rocprof.py $O/prof_2p20 "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 (2^20 block only; 100 timed + 100 profiled-pass proofs + warm-up), round 6" > $O/bench_2p20_kernel_stats.md 2>> $O/rocprof_2p20.err
rm -rf $O/prof_2p20
rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/$O/prof_plonk -- python3 tools/pmc_plonk.py 22 3 > $O/plonk_under_rocprof.json 2> $O/rocprof_plonk.err
python tools/summarize_rocprof.py $O/prof_plonk "rocprofv3 --kernel-trace --stats -- python3 tools/pmc_plonk.py 22 3 (plonk.Setup + 3 x 2 proofs at 2^22 gates), round 6" > $O/plonk_kernel_stats.md 2>> $O/rocprof_plonk.err
rm -rf $O/prof_plonk
B20="python3 bench.py --steps 2 --warmup 1 --no-2p24 --no-plonk --no-micro --no-export --no-cpu-baseline --no-host-inputs"
B24="python3 bench.py --log-n 24 --steps 2 --warmup 1 --no-plonk --no-micro --no-export --no-cpu-baseline --no-host-inputs"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $PWD/$O/pmc_f20 -- $B20 > /dev/null 2> $O/pmc_f20.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $PWD/$O/pmc_w20 -- $B20 > /dev/null 2> $O/pmc_w20.err
cp profiles/pmc_traffic.json $O/pmc_traffic.json
python tools/pmc_traffic.py $O/pmc_f20 $O/pmc_w20 $O/pmc_traffic.json 20 "round 6 HEAD" | tee $O/pmc_20.txt
rm -rf $O/pmc_f20 $O/pmc_w20
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $PWD/$O/pmc_f24 -- $B24 > /dev/null 2> $O/pmc_f24.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $PWD/$O/pmc_w24 -- $B24 > /dev/null 2> $O/pmc_w24.err
python tools/pmc_traffic.py $O/pmc_f24 $O/pmc_w24 $O/pmc_traffic.json 24 "round 6 HEAD" | tee $O/pmc_24.txt
rm -rf $O/pmc_f24 $O/pmc_w24
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $PWD/$O/pmc_fp -- python3 tools/pmc_plonk.py 22 1 > /dev/null 2> $O/pmc_fp.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $PWD/$O/pmc_wp -- python3 tools/pmc_plonk.py 22 1 > /dev/null 2> $O/pmc_wp.err
python tools/pmc_traffic.py $O/pmc_fp $O/pmc_wp $O/pmc_traffic.json plonk_22 "round 6 HEAD, PLONK alone (tools/pmc_plonk.py 22 1)" | tee $O/pmc_plonk.txt
rm -rf $O/pmc_fp $O/pmc_wp
ls -la $O | head -40
