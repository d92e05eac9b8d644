set -u
O=gpurun_out/r04r; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_plonk.py tests/test_gpu_goffi.py tests/test_gpu_multidev.py -m gpu -q -x > $O/pytest_plonk.txt 2>&1; tail -4 $O/pytest_plonk.txt
for b in 0 1; do for ln in 17 18 19 20 22; do
  ZKMI_PLONK_BATCH3=$b ZKMI_BENCH_PLONK_REPS=8 timeout 300 python bench.py --lib exp --steps 5 --no-2p24 --no-micro --no-export --no-cpu-baseline --no-host-inputs --plonk-log-n $ln > $O/p.json 2> $O/p.err
  python -c "
import json;b=json.load(open('$O/p.json'));k=[x for x in b if x.startswith('plonk_2p')][0];p=b[k];print(json.dumps({'batch3':$b,'log_n':$ln,'prove_ms':p['prove_ms'],'ok':p['proof_verifies'],'rounds':p['rounds_ms']}))" | tee -a $O/plonk_batch3.jsonl
done; done
