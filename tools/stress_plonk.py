#!/usr/bin/env python3
"""Stability check of the PLONK path: many back-to-back proofs of the same instance (commitments on concurrent host threads, transforms on the main
stream) must all be byte-identical; ordering bugs between the commit threads and the main stream would show up as rare mismatches.
Usage: stress_plonk.py [log_n=16] [count=200]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from noir_backend_using_gnark_amd import _lib, kzg, plonk as zp  # noqa: E402

L = _lib.lib()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n = 1 << log_n
npub, nvars = 3, n // 2
nc = n - npub
srs = kzg.new_srs(n + 3, np.array([5, 6, 7, 8], dtype=np.uint64))
rng = np.random.default_rng(1)
xa, xb, xc = (rng.integers(0, nvars, nc, dtype=np.uint32) for _ in range(3))
dsol = _lib.DeviceBuffer(nvars * 32)
_lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(dsol.ptr), C.c_size_t(nvars), C.c_uint64(0x51), C.c_int(1), C.c_int(1), None))
coef = []
for sd in (1, 2, 3, 4):
    b = _lib.DeviceBuffer(nc * 32)
    _lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(b.ptr), C.c_size_t(nc), C.c_uint64(sd), C.c_int(1), C.c_int(0), None))
    coef.append(b)
dqk = _lib.DeviceBuffer(nc * 32)
dx = [_lib.DeviceBuffer.from_numpy(v) for v in (xa, xb, xc)]
_lib.check(L.zk_bn254_plonk_synth_qk_dev(C.c_void_p(dqk.ptr), *[C.c_void_p(b.ptr) for b in coef], *[C.c_void_p(b.ptr) for b in dx], C.c_void_p(dsol.ptr), C.c_size_t(nc), None))
pk = zp.setup(zp.Circuit(npub, nvars, coef[0], coef[1], coef[2], coef[3], dqk, xa, xb, xc), srs.g1)
bl = np.arange(1, 37, dtype=np.uint64).reshape(9, 4)
first = zp.prove(pk, dsol, bl)
bad = sum(zp.prove(pk, dsol, bl) != first for _ in range(count))
print("PLONK 2^%d rows: %d proofs, %d mismatches, proof %s" % (log_n, count, bad, first.hex()[:16]))
sys.exit(1 if bad else 0)
