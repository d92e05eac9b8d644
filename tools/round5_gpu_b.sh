#!/bin/bash
# Round 5, batch B: the new Groth16 export path -- its tests, then the 2^20 worker (tools/export_bench_groth16.py), cold and warm.
set -u
O=gpurun_out/${1:-rnd5b}
mkdir -p $O /tmp/g16
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_keyio.py tests/test_gpu_goffi.py -m gpu -x -q --durations=8 > $O/pytest_keyio_goffi.txt 2>&1; echo "rc=$?" >> $O/pytest_keyio_goffi.txt; tail -25 $O/pytest_keyio_goffi.txt
python tools/export_bench_groth16.py make /tmp/g16 20 > $O/g16_make.json 2> $O/g16_make.err; cat $O/g16_make.json
timeout 900 python tools/export_bench_groth16.py preprocess /tmp/g16 > $O/g16_preprocess.json 2> $O/g16_preprocess.err; echo "preprocess rc=$?"; tail -c 2500 $O/g16_preprocess.json; tail -3 $O/g16_preprocess.err
timeout 900 python tools/export_bench_groth16.py prove /tmp/g16 10 > $O/g16_prove.json 2> $O/g16_prove.err; echo "prove rc=$?"; tail -c 3500 $O/g16_prove.json; tail -3 $O/g16_prove.err
timeout 300 python tools/export_bench_groth16.py verify /tmp/g16 > $O/g16_verify.json 2> $O/g16_verify.err; echo "verify rc=$?"; cat $O/g16_verify.json
