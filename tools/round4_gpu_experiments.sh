set -u
O=gpurun_out/r04p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_plonk.py tests/test_gpu_goffi.py -m gpu -q -x -k "not 2p20 and not 2p22 and not random_circuits" > $O/pytest_plonk.txt 2>&1; tail -3 $O/pytest_plonk.txt
for ln in 17 18 19 22; do
  ZKMI_BENCH_SRS_C=20 ZKMI_BENCH_PLONK_REPS=10 timeout 300 python bench.py --steps 5 --no-2p24 --no-micro --no-export --no-cpu-baseline --no-host-inputs --plonk-log-n $ln > $O/p.json 2> $O/p.err
  python -c "
import json;b=json.load(open('$O/p.json'));k=[x for x in b if x.startswith('plonk_2p')][0];p=b[k];print(json.dumps({'log_n':$ln,'prove_ms':p['prove_ms'],'ok':p['proof_verifies'],'rounds':p['rounds_ms']}))" | tee -a $O/plonk_host_muls_parallel.jsonl
done
