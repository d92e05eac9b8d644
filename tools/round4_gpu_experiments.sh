set -u
O=gpurun_out/r05o; mkdir -p $O
for sd in 0 1 0 1; do for ln in 19 20 22; do
  ZKMI_PLONK_COSET_SIDE=$sd ZKMI_BENCH_PLONK_REPS=6 timeout 600 python bench.py --lib exp --steps 5 --no-2p24 --no-micro --no-export --no-cpu-baseline --no-host-inputs --plonk-log-n $ln > $O/p.json 2> $O/p.err
  python -c "
import json;b=json.load(open('$O/p.json'));k=[x for x in b if x.startswith('plonk_2p')][0];p=b[k];print(json.dumps({'side':$sd,'log_n':$ln,'prove_ms':p['prove_ms'],'same':p['same_bytes_both_ways'],'ok':p['proof_verifies'],'rounds':p['rounds_ms']}))" | tee -a $O/plonk_coset_side.jsonl
done; done
