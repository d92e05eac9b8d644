#!/bin/bash
# round 3, final validation F: the driver's multi-GPU commands as dry runs (ranks share the one GPU over gloo; timings meaningless, proof bytes and cross-checks are the point)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3lf; mkdir -p $O
cd $R
export ZKMI_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1
for n in 2 4; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+n)) bench.py --gpus $n --steps 2 --warmup 1 --no-cpu-baseline > $O/dryrun_n$n.json 2> $O/dryrun_n$n.err
  grep "^{" $O/dryrun_n$n.json | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('N=%d' % d['n_gpus'], d['config']['workload'], d['proof_sha'], d.get('parity_error'), {k:(v.get('msm_same_on_every_rank'), v.get('msm_equals_odd_split_recombination'), v.get('ntt_inverse_of_forward_is_identity')) for k,v in d.items() if k.startswith('micro_')})"
done
python bench.py --log-n 21 --steps 2 --warmup 1 --no-cpu-baseline --no-host-inputs --no-2p24 --no-plonk --no-micro | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single 2^21', d['proof_sha'])"
python bench.py --log-n 22 --steps 2 --warmup 1 --no-cpu-baseline --no-host-inputs --no-2p24 --no-plonk --no-micro | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single 2^22', d['proof_sha'])"
python bench.py --log-n 23 --steps 2 --warmup 1 --no-cpu-baseline --no-host-inputs --no-2p24 --no-plonk --no-micro | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single 2^23', d['proof_sha'])"
