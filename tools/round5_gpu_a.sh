#!/bin/bash
# Round 5, batch A (baseline before this round's changes): the Groth16 export path at 2^20 constraints through libgnark_backend.so as round 4 left it
# (tools/export_bench_groth16.py: Preprocess / ProveWithPK / VerifyWithVK, one process per mode), and a kernel timeline of the witness-like 2^20 proof.
set -u
O=gpurun_out/${1:-rnd5a}
mkdir -p $O /tmp/g16
export TMPDIR=/tmp
python tools/export_bench_groth16.py make /tmp/g16 20 > $O/g16_make.json 2> $O/g16_make.err; cat $O/g16_make.json
timeout 900 python tools/export_bench_groth16.py preprocess /tmp/g16 > $O/g16_preprocess.json 2> $O/g16_preprocess.err; echo "preprocess rc=$?"; tail -c 1500 $O/g16_preprocess.json; tail -3 $O/g16_preprocess.err
timeout 900 python tools/export_bench_groth16.py prove /tmp/g16 4 > $O/g16_prove.json 2> $O/g16_prove.err; echo "prove rc=$?"; tail -c 2500 $O/g16_prove.json; tail -3 $O/g16_prove.err
timeout 300 python tools/export_bench_groth16.py verify /tmp/g16 > $O/g16_verify.json 2> $O/g16_verify.err; echo "verify rc=$?"; cat $O/g16_verify.json
ls -la /tmp/g16
# witness-like 2^20 timeline
rocprofv3 --kernel-trace --output-format csv -d $PWD/$O/prof_w -- python3 bench.py --steps 20 --scalars witness --no-2p24 --no-plonk --no-micro --no-export --no-cpu-baseline --no-host-inputs > $O/bench_witness_under_rocprof.json 2> $O/rocprof_w.err
python tools/timeline_proof.py $O/prof_w groth16 20 > $O/timeline_groth16_2p20_witness_like.txt 2>> $O/rocprof_w.err; tail -60 $O/timeline_groth16_2p20_witness_like.txt
rm -rf $O/prof_w
