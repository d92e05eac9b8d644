set -u
O=gpurun_out/r04j; mkdir -p $O
for ln in 16 17 18 19; do for c in 0 14 15 16 17 18 19 20; do
  ZKMI_BENCH_KEY_C=$c timeout 200 python bench.py --lib exp --steps 30 --log-n $ln --no-2p24 --no-plonk --no-micro --no-export --no-cpu-baseline --no-host-inputs > $O/g.json 2> $O/g.err
  python -c "
import json;b=json.load(open('$O/g.json'));km=b['roofline']['kernel_ms_per_step'];print(json.dumps({'groth16_log_n':$ln,'key_c':$c,'prove_ms':b['ms_per_step'],'acc_g1':km.get('msm_accumulate_g1'),'acc_g2':km.get('msm_accumulate_g2'),'fold_multi':km.get('msm_fold_multi'),'sha':b['proof_sha']}))" 2>/dev/null | tee -a $O/groth16_window_sweep.jsonl || tail -2 $O/g.err
done; done
