#!/bin/bash
# round 3, final validation D: the driver's default command on the final build (twice: boxes and runs scatter by 2-3 %), smoke
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3hd; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
python bench.py > $O/bench_default_2.json 2> $O/bench_default_2.err; tail -c 300 $O/bench_default_2.json
