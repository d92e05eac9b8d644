# scratch: the LAST one-off experiment of round 4 as it was sent to a GPU box (rewritten per run; results are copied to profiles/ by hand -- see profiles/INDEX.md).
# The repeatable measurement batch is tools/round4_gpu.sh.
set -u
O=gpurun_out/r06c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_goffi.py tests/test_gpu_plonk.py -m gpu -q -x -k "goffi or export or cache or handle_values" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for i in 1 2; do timeout 900 python bench.py --steps 3 --no-2p24 --no-plonk --no-micro --no-cpu-baseline --no-host-inputs > $O/b_$i.json 2> $O/b_$i.err
python -c "
import json;d=json.loads([l for l in open('$O/b_$i.json') if l.startswith('{')][-1]);e=d['export_path'];p=e['prove_process'];print(json.dumps({'warm':e['warm_PlonkProveWithPK_ms'],'prove':e['zk_bn254_plonk_prove_ms'],'ratio':e['warm_over_prove'],'warm_phases':p['warm_phases_per_call'],'cold':p['cold_PlonkProveWithPK_ms']}))"
done
