#!/bin/bash
# timelines of the accumulate chain: event-chained streams (default) against one shared stream (ZKMI_CHAIN=1)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3x; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export ZKMI_USE_EXPERIMENTS_LIB=1
for v in 0 1; do
ZKMI_CHAIN=$v rocprofv3 --kernel-trace --output-format csv -d $O/tr$v -o st -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-inputs --no-2p24 --no-plonk --no-micro > $O/b$v.json 2> $O/b$v.err
python3 $R/tools/timeline_proof.py $O/tr$v groth16 60 > $O/timeline_chain$v.txt 2>&1
rm -rf $O/tr$v
python3 -c "
import json; d=json.loads(open('$O/b$v.json').read().strip().splitlines()[-1]); print('CHAIN=$v', d['ms_per_step'])"
grep -E "k_accumulate|span" $O/timeline_chain$v.txt | cut -c1-100
done
