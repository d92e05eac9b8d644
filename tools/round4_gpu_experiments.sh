set -u
O=gpurun_out/r04w; mkdir -p $O
python tools/export_bench.py make /tmp/exp 19 > $O/make.json 2>&1
g++ -O2 -std=c++17 tools/lower_bench.cpp -lpthread -o /tmp/lower_bench
for i in 1 2 3; do /tmp/lower_bench /tmp/exp/acir.json 524282 skip; done | tee $O/lower_bench.jsonl
timeout 600 python -m pytest tests/test_gpu_goffi.py tests/test_gpu_plonk.py -m gpu -q -x -k "goffi or export or acir" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for i in 1 2; do timeout 900 python bench.py --steps 3 --no-2p24 --no-plonk --no-micro --no-cpu-baseline --no-host-inputs > $O/bench_export_$i.json 2> $O/bench_export_$i.err
python -c "
import json;d=json.loads([l for l in open('$O/bench_export_$i.json') if l.startswith('{')][-1]);e=d['export_path'];print(json.dumps({'pre':e['preprocess_process']['PlonkPreprocess_ms'],'pre_phases':e['preprocess_process']['phases'],'cold':e['prove_process']['cold_PlonkProveWithPK_ms'],'cold_phases':e['prove_process']['cold_phases'],'warm':e['warm_PlonkProveWithPK_ms'],'prove':e['zk_bn254_plonk_prove_ms'],'reader':e['acir_reader']}))"
done
