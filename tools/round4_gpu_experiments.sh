# scratch: the LAST one-off experiment of round 4 as it was sent to a GPU box (rewritten per run; results are copied to profiles/ by hand -- see profiles/INDEX.md).
# The repeatable measurement batch is tools/round4_gpu.sh.
set -u
O=gpurun_out/r06f; mkdir -p $O
for zs in 0 1 0 1; do for ln in 19 20 22; do
  ZKMI_PLONK_Z_SIDE=$zs ZKMI_BENCH_PLONK_REPS=6 timeout 600 python bench.py --lib exp --steps 5 --no-2p24 --no-micro --no-export --no-cpu-baseline --no-host-inputs --plonk-log-n $ln > $O/p.json 2> $O/p.err
  python -c "
import json;b=json.load(open('$O/p.json'));k=[x for x in b if x.startswith('plonk_2p')][0];p=b[k];print(json.dumps({'z_side':$zs,'log_n':$ln,'prove_ms':p['prove_ms'],'same':p['same_bytes_both_ways'],'ok':p['proof_verifies'],'rounds':p['rounds_ms']}))" | tee -a $O/plonk_z_side.jsonl
done; done
