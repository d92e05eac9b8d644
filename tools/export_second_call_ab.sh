#!/bin/bash
# Second proving call of a process (the one that queues window tables + high-priority streams on the background thread): how long it takes with the background
# jobs yielding to calls in flight (default 250 ms) and not yielding (0).   usage: tools/export_second_call_ab.sh OUTDIR [log2 constraints]
out=$1; lg=${2:-20}
d=$(mktemp -d)
python tools/export_bench_groth16.py make $d $lg > /dev/null
python tools/export_bench_groth16.py preprocess $d > /dev/null
for rep in 1 2 3; do
  for y in 250 0; do
    ZKMI_TOOL_BG_YIELD_MS=$y python tools/export_bench_groth16.py prove $d 10 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'yield_ms': $y, 'rep': $rep, 'cold_ms': d['cold_ProveWithPK_ms'], 'second_ms': d['second_ProveWithPK_ms'], 'background_after_second_ms': d['background_after_second_ms'], 'warm_ms': d['warm_ProveWithPK_ms'], 'second_prove_phase': d['second_phases'].get('groth16_prove'), 'bg': d['background_phases'], 'second_verifies': d['second_proof_verifies']}))" >> $out/export_second_call_ab.jsonl
  done
done
rm -rf $d
