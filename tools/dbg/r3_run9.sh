#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3i; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
grep -i -E "icache|ifetch|inst_cache|SQC_|instr.*fetch" $O/counters_list.txt | head -60 > $O/counters_icache.txt; cat $O/counters_icache.txt | cut -c1-200 | head -40
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH --output-format csv -d $O/pmc_ic -o c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-inputs --no-2p24 --no-plonk --no-micro > $O/pmc_ic.log 2>&1
tail -3 $O/pmc_ic.log | cut -c1-300
python3 $R/tools/pmc_summary.py $O/pmc_ic > $O/pmc_icache_summary.txt 2>&1; head -60 $O/pmc_icache_summary.txt
rm -rf $O/pmc_ic
