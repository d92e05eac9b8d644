#!/bin/bash
# Round 5, batch D: cold-path phases of the Groth16 exports at 2^20 (threaded rows, host-side tables timed), the new tests, zero-digit compaction A/B in
# alternating pairs (2^20 and 2^24, uniform), table widths 22 / 23 / 24 at 2^24.
set -u
O=gpurun_out/${1:-rnd5d}
mkdir -p $O /tmp/g16
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "prepared or window_bits or table_window" --durations=5 > $O/pytest_sel.txt 2>&1; echo "rc=$?" >> $O/pytest_sel.txt; tail -12 $O/pytest_sel.txt
python tools/export_bench_groth16.py make /tmp/g16 20 > $O/g16_make.json 2> $O/g16_make.err
timeout 900 python tools/export_bench_groth16.py preprocess /tmp/g16 > $O/g16_preprocess.json 2> $O/g16_preprocess.err; echo "preprocess rc=$?"; tail -c 2200 $O/g16_preprocess.json; tail -3 $O/g16_preprocess.err
timeout 900 python tools/export_bench_groth16.py prove /tmp/g16 10 > $O/g16_prove.json 2> $O/g16_prove.err; echo "prove rc=$?"; tail -c 3800 $O/g16_prove.json; tail -3 $O/g16_prove.err
timeout 300 python tools/export_bench_groth16.py verify /tmp/g16 > $O/g16_verify.json 2> $O/g16_verify.err; echo "verify rc=$?"; cat $O/g16_verify.json
g++ -O2 -std=c++17 tools/raw_lower_bench.cpp -lpthread -o /tmp/raw_lower_bench && /tmp/raw_lower_bench /tmp/g16/raw.json | tee $O/raw_lower_bench.json
make -C noir_backend_using_gnark_amd/csrc EXPERIMENTS=1 -j16 > $O/make_exp.log 2>&1; echo "make exp rc=$?"
python tools/ab_bench.py $O/drop_zero_digits_pairs.jsonl --steps 100 --only 2p20 "drop1:ZKMI_W_DROP_ZERO_DIGITS=1" "drop0:ZKMI_W_DROP_ZERO_DIGITS=0" "drop1:ZKMI_W_DROP_ZERO_DIGITS=1" "drop0:ZKMI_W_DROP_ZERO_DIGITS=0" "drop1:ZKMI_W_DROP_ZERO_DIGITS=1" "drop0:ZKMI_W_DROP_ZERO_DIGITS=0" 2>&1 | cut -c1-400
python tools/ab_bench.py $O/table_width_2p24.jsonl --steps 20 --only 2p24 "c22_drop1:ZKMI_BENCH_KEY_C=22" "c24:ZKMI_BENCH_KEY_C=24" "c23:ZKMI_BENCH_KEY_C=23" "c22_drop0:ZKMI_BENCH_KEY_C=22,ZKMI_W_DROP_ZERO_DIGITS=0" "c24:ZKMI_BENCH_KEY_C=24" 2>&1 | cut -c1-500
