set -u
O=gpurun_out/r05g; mkdir -p $O
run() { timeout 300 python bench.py $1 --steps 60 --no-2p24 --no-plonk --no-micro --no-export --no-cpu-baseline --no-host-inputs > $O/p.json 2> $O/p.err
  python -c "
import json;d=json.loads([l for l in open('$O/p.json') if l.startswith('{')][-1]);print(json.dumps({'cfg':'$2','ms':d['ms_per_step'],'sha':d['proof_sha']}))" | tee -a $O/alias.jsonl; }
run "--lib exp" base
for q in 4 16; do export GPU_MAX_HW_QUEUES=$q
for a in 101 102 103 104 112 113 114 123 124 134; do ZKMI_ALIAS_LO=$a run "--lib exp" hwq${q}_lo_$a; done
for a in 101 102 104 112 114 124; do ZKMI_ALIAS_HI=$a run "--lib exp" hwq${q}_hi_$a; done
done
unset GPU_MAX_HW_QUEUES
run "--lib exp" base_again
