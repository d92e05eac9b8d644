#!/usr/bin/env python3
"""Stability check: many back-to-back proofs of the same instance must all be byte-identical (stream / event ordering bugs would show up
as rare mismatches).  Usage: stress_prove.py [log_n] [count]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import noir_backend_using_gnark_amd as zk  # noqa: E402
from noir_backend_using_gnark_amd import _lib  # noqa: E402

L = _lib.lib()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 18
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
N = 1 << log_n
npub = 8


def g1(seed, n):
    b = _lib.DeviceBuffer(n * 64)
    _lib.check(L.zk_bn254_g1_generate_dev(C.c_void_p(b.ptr), C.c_size_t(n), C.c_uint64(seed), None))
    return b


def g2(seed, n):
    b = _lib.DeviceBuffer(n * 128)
    _lib.check(L.zk_bn254_g2_generate_dev(C.c_void_p(b.ptr), C.c_size_t(n), C.c_uint64(seed), None))
    return b


def fr(seed, n, wit=0):
    b = _lib.DeviceBuffer(n * 32)
    _lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(b.ptr), C.c_size_t(n), C.c_uint64(seed), C.c_int(1), C.c_int(wit), None))
    return b


a, bq, k, z, b2 = g1(1, N), g1(2, N), g1(3, N), g1(4, N), g2(5, N)
small = [g1(s, 1).to_numpy(np.uint64, (8,)) for s in (6, 7, 8)]
small2 = [g2(s, 1).to_numpy(np.uint64, (16,)) for s in (9, 10)]
pk = zk.ProvingKey(log_n, N, npub, small[0], small[1], small[2], a, bq, k.ptr + npub * 64, z, small2[0], small2[1], b2, bases_on_device=True)
da, db = fr(11, N), fr(12, N)
dc = _lib.DeviceBuffer(N * 32)
_lib.check(L.zk_bn254_fr_mul_dev(C.c_void_p(dc.ptr), C.c_void_p(da.ptr), C.c_void_p(db.ptr), C.c_size_t(N), None))
rs = fr(13, 2).to_numpy(np.uint64, (2, 4))
bad = 0
for wit in (0, 1):
    dw = fr(14, N, wit)
    first = zk.prove(pk, da, db, dc, dw, rs[0], rs[1], n_constraints=N, on_device=True)
    for i in range(count):
        if zk.prove(pk, da, db, dc, dw, rs[0], rs[1], n_constraints=N, on_device=True) != first:
            bad += 1
    print("2^%d, %s scalars: %d proofs, %d mismatches, proof %s" % (log_n, "witness-like" if wit else "uniform", count, bad, first.hex()[:16]))
sys.exit(1 if bad else 0)
