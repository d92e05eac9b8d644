set -u
O=gpurun_out/r05s; mkdir -p $O
for q in 4 2 8; do
  GPU_MAX_HW_QUEUES=$q ZKMI_BENCH_PLONK_REPS=4 timeout 900 python bench.py --steps 20 --no-micro --no-export --no-cpu-baseline --no-host-inputs > $O/p.json 2> $O/p.err
  python -c "
import json;d=json.loads([l for l in open('$O/p.json') if l.startswith('{')][-1]);print(json.dumps({'GPU_MAX_HW_QUEUES':$q,'groth16_2p20_ms':d['ms_per_step'],'groth16_2p24_ms':d['at_2p24']['prove_ms'],'plonk_2p22_ms':d['plonk_2p22']['prove_ms'],'plonk_2p22_coeff_ms':d['plonk_2p22']['prove_ms_lro_from_coefficients']}))" | tee -a $O/hwq_all_sizes.jsonl
done
