set -u
O=gpurun_out/r05r; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_plonk.py tests/test_gpu_goffi.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for ln in 14 17 19 20 22; do
  ZKMI_BENCH_PLONK_REPS=6 timeout 600 python bench.py --steps 5 --no-2p24 --no-micro --no-export --no-cpu-baseline --no-host-inputs --plonk-log-n $ln > $O/p.json 2> $O/p.err
  python -c "
import json;b=json.load(open('$O/p.json'));k=[x for x in b if x.startswith('plonk_2p')][0];p=b[k];print(json.dumps({'log_n':$ln,'prove_ms':p['prove_ms'],'coeff_ms':p['prove_ms_lro_from_coefficients'],'same':p['same_bytes_both_ways'],'ok':p['proof_verifies'],'rounds':p['rounds_ms']}))" | tee -a $O/plonk_host_straus.jsonl
done
