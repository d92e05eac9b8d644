#!/usr/bin/env python3
"""Per-kernel timing breakdown (hipEvent pairs inside libzkmi) for one MSM / NTT / computeH / prove at a given size."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import noir_backend_using_gnark_amd as zk  # noqa: E402
MONT = zk.MultiExpConfig(scalars_mont=True)
from noir_backend_using_gnark_amd import _lib  # noqa: E402
from noir_backend_using_gnark_amd import bn254 as zb  # noqa: E402

L = _lib.lib()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
which = sys.argv[2] if len(sys.argv) > 2 else "all"


def show(title, wall_ms):
    prof = _lib.profile_read()
    tot = sum(v[1] for v in prof.values())
    print("== %s: wall %.3f ms, sum of kernels %.3f ms" % (title, wall_ms, tot))
    for k, (cnt, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        print("   %-28s x%-4d %9.3f ms  (%5.1f%%)" % (k, cnt, ms, 100 * ms / tot if tot else 0))
    _lib.profile_reset()


t0 = time.time()
dp, ds = _lib.DeviceBuffer(n * 64), _lib.DeviceBuffer(n * 32)
_lib.check(L.zk_bn254_g1_generate_dev(C.c_void_p(dp.ptr), C.c_size_t(n), C.c_uint64(0xB1), None))
print("g1 generate 2^%d: %.1f ms" % (log_n, 1e3 * (time.time() - t0)))
for witness in (0, 1):
    _lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(ds.ptr), C.c_size_t(n), C.c_uint64(0xC), C.c_int(1), C.c_int(witness), None))
    for c in ([0] if which != "sweep" else [12, 14, 15, 16, 17, 18, 20]):
        cfg = zk.MultiExpConfig(scalars_mont=True, window_bits=c)
        zb.g1_multi_exp_dev(dp.ptr, ds.ptr, n, cfg)  # warm
        _lib.profile(False)
        t0 = time.time()
        for _ in range(3):
            zb.g1_multi_exp_dev(dp.ptr, ds.ptr, n, cfg)
        wall = (time.time() - t0) / 3 * 1e3
        _lib.profile(True)
        _lib.profile_reset()
        zb.g1_multi_exp_dev(dp.ptr, ds.ptr, n, cfg)
        show("G1 MSM 2^%d %s c=%d  (%.1f M scalar-mul/s unprofiled)" % (log_n, "witness-like" if witness else "uniform", c, n / wall / 1e3), wall)
        _lib.profile(False)

# NTT
dom = zk.Domain(n)
dom.fft(ds, zk.DIF)
t0 = time.time()
for _ in range(5):
    dom.fft(ds, zk.DIF)
wall = (time.time() - t0) / 5 * 1e3
_lib.profile(True)
_lib.profile_reset()
dom.fft(ds, zk.DIF)
show("NTT 2^%d DIF (%.2f G elem/s unprofiled)" % (log_n, n / wall / 1e6), wall)
_lib.profile(False)

# computeH
da, db, dc, dh = (_lib.DeviceBuffer(n * 32) for _ in range(4))
for i, d in enumerate((da, db, dc)):
    _lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(d.ptr), C.c_size_t(n), C.c_uint64(10 + i), C.c_int(1), C.c_int(0), None))
_lib.check(L.zk_bn254_groth16_compute_h_dev(C.c_void_p(da.ptr), C.c_void_p(db.ptr), C.c_void_p(dc.ptr), C.c_size_t(n), C.c_uint32(log_n), C.c_void_p(dh.ptr), None))
t0 = time.time()
_lib.check(L.zk_bn254_groth16_compute_h_dev(C.c_void_p(da.ptr), C.c_void_p(db.ptr), C.c_void_p(dc.ptr), C.c_size_t(n), C.c_uint32(log_n), C.c_void_p(dh.ptr), None))
wall = (time.time() - t0) * 1e3
_lib.profile(True)
_lib.profile_reset()
_lib.check(L.zk_bn254_groth16_compute_h_dev(C.c_void_p(da.ptr), C.c_void_p(db.ptr), C.c_void_p(dc.ptr), C.c_size_t(n), C.c_uint32(log_n), C.c_void_p(dh.ptr), None))
show("computeH 2^%d" % log_n, wall)
_lib.profile(False)

if which in ("all", "g2"):
    m = n // 4
    d2 = _lib.DeviceBuffer(m * 128)
    t0 = time.time()
    _lib.check(L.zk_bn254_g2_generate_dev(C.c_void_p(d2.ptr), C.c_size_t(m), C.c_uint64(0xB2), None))
    print("g2 generate %d: %.1f ms" % (m, 1e3 * (time.time() - t0)))
    _lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(ds.ptr), C.c_size_t(n), C.c_uint64(0xC), C.c_int(1), C.c_int(0), None))
    zb.g2_multi_exp_dev(d2.ptr, ds.ptr, m, config=MONT)
    t0 = time.time()
    zb.g2_multi_exp_dev(d2.ptr, ds.ptr, m, config=MONT)
    wall = (time.time() - t0) * 1e3
    _lib.profile(True)
    _lib.profile_reset()
    zb.g2_multi_exp_dev(d2.ptr, ds.ptr, m, config=MONT)
    show("G2 MSM %d uniform (%.1f M scalar-mul/s unprofiled)" % (m, m / wall / 1e3), wall)
    _lib.profile(False)
