#!/bin/bash
# The measurement batch of round 4 on one MI355X box (gpurun): full -m gpu suite, the default bench line, rocprofv3 kernel stats of the 2^20 workload and of the
# PLONK block, the single-process multi-entry bench (virtual entries on a one-GPU box), gloo dry runs of the multi-process path at a FIXED 2^24 (strong scaling).
# Everything lands under gpurun_out/$1/ ; the summaries judged are copied to profiles/ by hand.
set -u
O=gpurun_out/${1:-r04f}
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/pytest_full.txt 2>&1; echo "rc=$?" >> $O/pytest_full.txt
tail -25 $O/pytest_full.txt
timeout 900 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err; echo "bench rc=$?"
# rocprofv3: the 2^20 workload alone, then the PLONK block alone (the program itself after --)
rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/$O/prof_2p20 -- python3 bench.py --steps 100 --no-2p24 --no-plonk --no-micro --no-export --no-cpu-baseline --no-host-inputs > $O/bench_2p20_under_rocprof.json 2> $O/rocprof_2p20.err
python tools/summarize_rocprof.py $O/prof_2p20 "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 (2^20 block only), round 4" > $O/bench_2p20_kernel_stats.md 2>> $O/rocprof_2p20.err
rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/$O/prof_plonk -- python3 bench.py --steps 5 --no-2p24 --no-micro --no-export --no-cpu-baseline --no-host-inputs > $O/bench_plonk_under_rocprof.json 2> $O/rocprof_plonk.err
python tools/summarize_rocprof.py $O/prof_plonk "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 (2^20 + PLONK 2^22 blocks), round 4" > $O/bench_plonk_kernel_stats.md 2>> $O/rocprof_plonk.err
rm -rf $O/prof_2p20 $O/prof_plonk   # the traces are large; the summaries stay
# one process, several device entries (virtual on this box)
for n in 2 4 8; do timeout 600 python bench.py --gpus $n --single-process --steps 5 --warmup 2 --log-n 20 > $O/bench_single_process_n$n.json 2> $O/bench_single_process_n$n.err; done
timeout 900 python bench.py --gpus 2 --single-process --steps 3 --warmup 1 --log-n 22 > $O/bench_single_process_n2_2p22.json 2> $O/bench_single_process_n2_2p22.err
# the multi-process path at a fixed 2^24 (north_star: "2^24 on 1/2/4/8"): ranks share this GPU over gloo -- timings meaningless, bytes must be the 2^24 proof's
for cfg in "2 23" "4 22" "8 21"; do set -- $cfg; ZKMI_DIST_BACKEND=gloo timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port 2951$1 bench.py --gpus $1 --steps 1 --warmup 1 --log-n $2 --no-micro > $O/bench_gloo_dryrun_2p24_n$1.json 2> $O/bench_gloo_dryrun_2p24_n$1.err; echo "dry run n=$1 rc=$?"; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], d.get("n_gpus"), d.get("ms_per_step"), d.get("proof_sha"), d.get("proof_equals_single_entry"), d.get("parity_error"))
    except Exception as e:
        print(f, "unreadable", e)
PY
