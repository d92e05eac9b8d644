"""CPU suite: the N>1 path (range sharding + all-gather of partial sums + host combine) on 2 gloo processes.
The local partial sums are produced by the ORACLE here (no GPU in this container); what is under test is the product's
sharding arithmetic, the collective plumbing and the host-side combine (zk_bn254_g1_sum_xyzz, libzkmi)."""
import os
import subprocess
import sys

import numpy as np

from noir_backend_using_gnark_amd import parallel as par

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]

WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from noir_backend_using_gnark_amd import parallel as par, bn254 as zb
from oracle import oracle as orc, bn254_ref as ref
rank, world, _ = par.init_distributed("gloo")
assert world == 2
n = 1001
pts, sc = orc.g1_gen_points(5, n), orc.rand_fr(6, n)
lo, hi = par.shard_range(n, rank, world)
# local partial (affine from the oracle, lifted to XYZZ with ZZ = ZZZ = 1)
loc = orc.g1_msm(pts[lo:hi], sc[lo:hi])
one = np.frombuffer(ref.limbs_le(ref.to_mont(1, ref.Q)), dtype=np.uint64)
rec = np.concatenate([loc, one, one])
gathered = par.all_gather_limbs(rec)
assert gathered.shape == (2, 16)
total = zb.g1_sum_partials(gathered)
assert (total == orc.g1_msm(pts, sc)).all()
assert (gathered[rank] == rec).all()
# ---- window (table-row) sharding: every rank takes ALL points but only the digit windows w = rank, rank + world, ...; its partial sum is the
# MSM over the scalars restricted to those windows (gnark's signed c-bit recoding), and the partial sums add up to the whole MSM
import torch
c = 11
Wd = (255 + c - 1) // c
rows = par.window_rows(c, rank, world)
assert rows == list(range(rank, Wd, world))
half = 1 << (c - 1)
mine = []
for s in orc.from_mont_vec(sc):
    carry, acc = 0, 0
    for w in range(Wd):
        d = ((s >> (w * c)) & ((1 << c) - 1)) + carry
        carry = 0
        if d > half:
            d -= 1 << c
            carry = 1
        if w in rows:
            acc += d << (w * c)
    mine.append(acc %% ref.R)
loc = orc.g1_msm(pts, orc.to_mont_vec(mine))
gathered = par.all_gather_limbs(np.concatenate([loc, one, one]))
assert (zb.g1_sum_partials(gathered) == orc.g1_msm(pts, sc)).all()
# the coefficient exchange of that mode: blocks of h concatenated in rank order on every rank
blk = torch.arange(8, dtype=torch.int64).reshape(2, 4) + 100 * rank
full = par.all_gather_blocks(blk)
assert full.shape == (4, 4) and (full[:2] == torch.arange(8).reshape(2, 4)).all() and (full[2:] == torch.arange(8).reshape(2, 4) + 100).all()
par.dist().barrier()
sys.stdout.write("rank %%d ok\n" %% rank); sys.stdout.flush()
''' % ROOT


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 1 << 20, (1 << 20) - 1):
        for world in (1, 2, 3, 8):
            r = [par.shard_range(n, g, world) for g in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_all_gather_single_process():
    x = np.arange(16, dtype=np.uint64)
    assert (par.all_gather_limbs(x) == x.reshape(1, 16)).all()


def test_window_rows_partition_the_windows():
    for c in (9, 16, 20, 22):
        Wd = (255 + c - 1) // c
        for world in (1, 2, 4, 8):
            rows = [par.window_rows(c, g, world) for g in range(world)]
            assert sorted(sum(rows, [])) == list(range(Wd))
            assert max(map(len, rows)) - min(map(len, rows)) <= 1


def test_two_rank_sharded_msm_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rank 0 ok" in out.stdout and "rank 1 ok" in out.stdout
