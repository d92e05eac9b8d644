set -u
O=gpurun_out/r05d; mkdir -p $O
run() { timeout 300 python bench.py $1 --steps 100 --no-2p24 --no-plonk --no-micro --no-export --no-cpu-baseline --no-host-inputs > $O/p.json 2> $O/p.err
  python -c "
import json;d=json.loads([l for l in open('$O/p.json') if l.startswith('{')][-1]);print(json.dumps({'cfg':'$2','ms':d['ms_per_step'],'sha':d['proof_sha'],'k':{k:v for k,v in list(d['roofline']['kernel_ms_per_step'].items())[:8]}}))" | tee -a $O/product.jsonl; }
run "" product; run "" product
ZKMI_INIT_STREAMS=2 run "--lib exp" init_2
for i in 1 2; do timeout 900 python bench.py --steps 3 --no-2p24 --no-micro --no-cpu-baseline --no-host-inputs > $O/bench_export_$i.json 2> $O/bench_export_$i.err
python -c "
import json;d=json.loads([l for l in open('$O/bench_export_$i.json') if l.startswith('{')][-1]);e=d['export_path'];p=e['prove_process'];print(json.dumps({'plonk22':d['plonk_2p22']['prove_ms'],'pre':e['preprocess_process']['PlonkPreprocess_ms'],'cold':p['cold_PlonkProveWithPK_ms'],'cold_phases':p['cold_phases'],'second':p['second_PlonkProveWithPK_ms'],'warm':e['warm_PlonkProveWithPK_ms'],'prove':e['zk_bn254_plonk_prove_ms'],'verify':e['verify_process']['cold_PlonkVerifyWithVK_ms']}))"
done
