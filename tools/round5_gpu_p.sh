#!/bin/bash
# Round 5, batch P: how a fresh process best moves 0.37 GB of pageable key text into HBM (tools/h2d_bench.hip; one variant per process)
set -u
O=gpurun_out/${1:-rnd5p}
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for v in "plain" "two" "register" "staged 4" "staged 8" "staged 16" "decode 8" "decode 16" "decode 32"; do
  timeout 120 tools/h2d_bench $v >> $O/h2d_bench.jsonl 2>> $O/h2d_bench.err
done
done
cat $O/h2d_bench.jsonl
