set -u
O=gpurun_out/r05f; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_goffi.py tests/test_gpu_plonk.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for i in 1 2 3; do timeout 900 python bench.py --steps 3 --no-2p24 --no-plonk --no-micro --no-cpu-baseline --no-host-inputs > $O/bench_export_$i.json 2> $O/bench_export_$i.err
python -c "
import json;d=json.loads([l for l in open('$O/bench_export_$i.json') if l.startswith('{')][-1]);e=d['export_path'];p=e['prove_process'];print(json.dumps({'pre':e['preprocess_process']['PlonkPreprocess_ms'],'pre_phases':e['preprocess_process']['phases'],'cold':p['cold_PlonkProveWithPK_ms'],'second':p['second_PlonkProveWithPK_ms'],'warm':e['warm_PlonkProveWithPK_ms'],'prove':e['zk_bn254_plonk_prove_ms'],'verify':e['verify_process']}))"
done
