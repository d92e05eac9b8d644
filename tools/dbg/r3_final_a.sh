#!/bin/bash
# round 3, final validation A (second pass, after the computeH launches were merged): the whole GPU suite, smoke, the default bench line
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3la; mkdir -p $O
cd $R
( time python -m pytest tests -q -m gpu -x ) > $O/gpu_pytest.log 2>&1; tail -6 $O/gpu_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
