#!/bin/bash
# Round 5, batch M: SQ counters of the 2^20 block (one pass, nothing but --pmc): how the waves of k_accumulate spend their cycles -- issuing, stalled at issue, parked.
set -u
O=gpurun_out/${1:-rnd5m}
mkdir -p $O
export TMPDIR=/tmp
B20="python3 bench.py --steps 2 --warmup 1 --no-2p24 --no-plonk --no-micro --no-export --no-cpu-baseline --no-host-inputs"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $PWD/$O/pmc_sq -- $B20 > /dev/null 2> $O/pmc_sq.err
python tools/pmc_summary.py $O/pmc_sq > $O/pmc_sq_2p20.txt 2>> $O/pmc_sq.err; head -14 $O/pmc_sq_2p20.txt
rm -rf $O/pmc_sq
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $PWD/$O/pmc_sq2 -- $B20 > /dev/null 2> $O/pmc_sq2.err
python tools/pmc_summary.py $O/pmc_sq2 > $O/pmc_sq2_2p20.txt 2>> $O/pmc_sq2.err; head -8 $O/pmc_sq2_2p20.txt; tail -3 $O/pmc_sq2.err
rm -rf $O/pmc_sq2
