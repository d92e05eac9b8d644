"""CPU suite: the block-sharded computeH schedule (parallel.compute_h_sharded: 4 compute phases, 10 transposes) run on the
oracle's big-integer arithmetic -- in lock-step over virtual ranks, and on 2 gloo processes through the product's own
exchange code.  The HIP phases are compared with the same schedule in tests/test_gpu_parity.py."""
import os
import subprocess
import sys

import pytest

from oracle import bn254_ref as ref
from tests import sharded_h_ref as sh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def test_int_transforms_match_the_oracle_domain():
    dom = ref.Domain(16)
    x = ref.rand_felts(9, 16)
    a = list(x)
    sh.dif_inplace(a, dom.gen)
    assert a == dom.fft(x, ref.DIF)
    a = list(x)
    sh.dit_inplace(a, dom.gen)
    assert a == dom.fft(x, ref.DIT)


@pytest.mark.parametrize("log_d,G", [(6, 1), (6, 2), (6, 4), (6, 8), (7, 8), (4, 4)])
def test_sharded_schedule_equals_compute_h(log_d, G):
    D = 1 << log_d
    M = D // G
    a, b, c = (ref.rand_felts(s, D) for s in (31, 32, 33))
    want = ref.compute_h(a, b, c, ref.Domain(D))
    blk = lambda v: [list(v[r * M:(r + 1) * M]) for r in range(G)]
    H = sh.run_virtual(sh.phase_int, blk(a), blk(b), blk(c), log_d)
    assert [x for h in H for x in h] == want
    H = sh.run_virtual(sh.phase_int, blk(a), blk(b), blk(c), log_d, per_array=True)
    assert [x for h in H for x in h] == want
    H = sh.run_virtual_six(sh.phase_int, blk(a), blk(b), blk(c), log_d)   # c in coefficient form: 9 transposes, same h for ANY a, b, c
    assert [x for h in H for x in h] == want


WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch
from noir_backend_using_gnark_amd import parallel as par
from oracle import bn254_ref as ref
from tests import sharded_h_ref as sh
rank, world, _ = par.init_distributed("gloo")
assert world == 2
log_d = 6
D = 1 << log_d; M = D // world
a, b, c = (ref.rand_felts(s, D) for s in (41, 42, 43))
want = ref.compute_h(a, b, c, ref.Domain(D))

def to_t(v):   # integers -> (M, 4) int64 limbs
    return torch.from_numpy(np.frombuffer(b"".join(ref.limbs_le(x) for x in v), dtype=np.int64).reshape(-1, 4).copy())
def to_i(t):
    raw = t.numpy().view(np.uint64)
    return [sum(int(raw[i, k]) << (64 * k) for k in range(4)) for i in range(raw.shape[0])]

def phase(p, ta, tb, tc, log_d, log_g, rk):
    la, lb, lc = to_i(ta), (to_i(tb) if tb is not None else None), (to_i(tc) if tc is not None else None)
    sh.phase_int(p, la, lb, lc, log_d, log_g, rk)
    for t, l in ((ta, la), (tb, lb), (tc, lc)):
        if t is not None: t.copy_(to_t(l))

blk = lambda v: to_t(v[rank * M:(rank + 1) * M])
h = par.compute_h_sharded(blk(a), blk(b), blk(c), log_d, rank, world, phase=phase)   # default: the six-transform schedule
assert to_i(h) == want[rank * M:(rank + 1) * M], "rank %%d block of h differs (six transforms)" %% rank
h = par.compute_h_sharded(blk(a), blk(b), blk(c), log_d, rank, world, phase=phase, six_transforms=False)
assert to_i(h) == want[rank * M:(rank + 1) * M], "rank %%d block of h differs" %% rank
# the pipelined schedule (per-array phases 0, 1, 4, then 5; transposes one array ahead) gives the same block
h = par.compute_h_sharded(blk(a), blk(b), blk(c), log_d, rank, world, phase=phase, pipelined=True, six_transforms=False)
assert to_i(h) == want[rank * M:(rank + 1) * M], "rank %%d block of h differs (pipelined)" %% rank
# the standalone sharded transform through the product's driver and exchange code: all eight modes
def nstep(st, t, log_d, log_g, rk, inverse, dec, coset):
    l = to_i(t)
    sh.ntt_step_int(st, l, log_d, log_g, rk, inverse, dec, coset)
    t.copy_(to_t(l))
dom = ref.Domain(D)
for inverse in (False, True):
    for dec in (ref.DIT, ref.DIF):
        for coset in (False, True):
            want_n = (dom.fft_inverse if inverse else dom.fft)(a, dec, coset)
            got = par.ntt_sharded(blk(a), log_d, rank, world, inverse, dec, coset, step=nstep)
            assert to_i(got) == want_n[rank * M:(rank + 1) * M], (rank, inverse, dec, coset)
# exchange is an involution
x = torch.arange(M * 4, dtype=torch.int64).reshape(M, 4) + 1000 * rank
assert torch.equal(par.block_exchange(par.block_exchange(x)), x)
par.dist().barrier()
sys.stdout.write("rank %%d ok\n" %% rank); sys.stdout.flush()
''' % ROOT


def test_two_rank_sharded_compute_h_gloo(tmp_path):
    script = tmp_path / "worker_h.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rank 0 ok" in out.stdout and "rank 1 ok" in out.stdout


@pytest.mark.parametrize("log_d,G", [(6, 1), (6, 2), (6, 4), (7, 8), (4, 4)])
def test_sharded_standalone_ntt_schedule_all_modes(log_d, G):
    """parallel.ntt_sharded's schedule (steps + two transposes; include/zkmi.h zk_bn254_ntt_shard_dev) on big integers over virtual ranks equals the
    oracle's (*Domain).FFT / FFTInverse for all eight mode combinations -- and the product's own driver runs the same sequence (checked by playing it
    for one rank at G = 1, where the exchanges are the identity)."""
    from noir_backend_using_gnark_amd import parallel as par
    D = 1 << log_d
    M = D // G
    dom = ref.Domain(D)
    x = ref.rand_felts(0x77 + log_d, D)
    for inverse in (False, True):
        for dec in (ref.DIT, ref.DIF):
            for coset in (False, True):
                want = (dom.fft_inverse if inverse else dom.fft)(x, dec, coset)
                blocks = [list(x[r * M:(r + 1) * M]) for r in range(G)]
                out = sh.run_virtual_ntt(sh.ntt_step_int, blocks, log_d, inverse, dec, coset)
                assert [v for b in out for v in b] == want, (inverse, dec, coset)
                if G == 1:
                    a = list(x)
                    got = par.ntt_sharded(a, log_d, 0, 1, inverse, dec, coset, step=sh.ntt_step_int, exchange=lambda v: v)
                    assert got == want
