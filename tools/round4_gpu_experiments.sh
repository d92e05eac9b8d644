set -u
O=gpurun_out/r05j; mkdir -p $O
for ln in 12 14 16 17 18 19 20 21 22; do
  ZKMI_BENCH_PLONK_REPS=6 timeout 600 python bench.py --steps 5 --no-2p24 --no-micro --no-export --no-cpu-baseline --no-host-inputs --plonk-log-n $ln > $O/p.json 2> $O/p.err
  python -c "
import json;b=json.load(open('$O/p.json'));k=[x for x in b if x.startswith('plonk_2p')][0];p=b[k];print(json.dumps({'log_n':$ln,'prove_ms':p['prove_ms'],'coeff_ms':p['prove_ms_lro_from_coefficients'],'lagrange_build_ms':p['lagrange_srs_ms_once_per_key'],'same':p['same_bytes_both_ways'],'ok':p['proof_verifies'],'round1':p['rounds_ms']['round1_lro_committed']}))" | tee -a $O/plonk_lagrange.jsonl
done
