set -u
O=gpurun_out/r06a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_goffi.py tests/test_gpu_multidev.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for i in 1 2; do timeout 900 python bench.py --steps 100 --no-2p24 --no-plonk --no-micro --no-cpu-baseline --no-host-inputs > $O/b_$i.json 2> $O/b_$i.err
python -c "
import json;d=json.loads([l for l in open('$O/b_$i.json') if l.startswith('{')][-1]);e=d['export_path'];p=e['prove_process'];print(json.dumps({'ms':d['ms_per_step'],'pre':e['preprocess_process']['PlonkPreprocess_ms'],'cold':p['cold_PlonkProveWithPK_ms'],'hip_init':p['cold_phases']['hip_init'],'warm':e['warm_PlonkProveWithPK_ms'],'verify':e['verify_process']['cold_PlonkVerifyWithVK_ms']}))"
done
for n in 2 4 8; do timeout 600 python bench.py --gpus $n --single-process --steps 5 --warmup 2 --log-n 20 > $O/sp_$n.json 2> $O/sp_$n.err; python -c "
import json;d=json.loads([l for l in open('$O/sp_$n.json') if l.startswith('{')][-1]);print('single process',$n,d['ms_per_step'],d.get('proof_equals_single_entry'))"; done
