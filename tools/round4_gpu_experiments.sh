set -u
O=gpurun_out/r05v; mkdir -p $O
for ln in 8 10 11; do for c in 0 8 10 12; do
  ZKMI_BENCH_SRS_C=$c ZKMI_BENCH_PLONK_REPS=10 timeout 300 python bench.py --steps 3 --no-2p24 --no-micro --no-export --no-cpu-baseline --no-host-inputs --plonk-log-n $ln > $O/p.json 2> $O/p.err
  python -c "
import json;b=json.load(open('$O/p.json'));k=[x for x in b if x.startswith('plonk_2p')][0];p=b[k];print(json.dumps({'log_n':$ln,'srs_table_c':$c,'prove_ms':p['prove_ms'],'coeff_ms':p['prove_ms_lro_from_coefficients'],'ok':p['proof_verifies'],'rounds':p['rounds_ms']}))" | tee -a $O/plonk_tiny_tables.jsonl
done; done
