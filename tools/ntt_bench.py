#!/usr/bin/env python3
"""NTT timing per size (data resident in HBM): ms, elements/s, algorithmic GB/s (64 B per element per transform).  usage: ntt_bench.py 20 22 24 26"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noir_backend_using_gnark_amd as zk  # noqa: E402
from noir_backend_using_gnark_amd import _lib  # noqa: E402

L = _lib.lib()
out = {}
for log_n in [int(a) for a in sys.argv[1:]] or [20, 24]:
    n = 1 << log_n
    d = _lib.DeviceBuffer(n * 32)
    _lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(d.ptr), C.c_size_t(n), C.c_uint64(7), C.c_int(1), C.c_int(0), None))
    dom = zk.Domain(n)
    dom.fft(d, zk.DIF)
    _lib.check(L.zk_dev_sync())
    reps = 50 if log_n <= 22 else 10
    t0 = time.perf_counter()
    for _ in range(reps):
        _lib.check(L.zk_bn254_ntt_dev(C.c_void_p(d.ptr), C.c_uint32(log_n), C.c_int(0), C.c_int(zk.DIF), C.c_int(0), C.c_void_p(0)))
    _lib.check(L.zk_dev_sync())
    dt = (time.perf_counter() - t0) / reps
    out[str(log_n)] = {"ms": round(dt * 1e3, 4), "elements_per_s": round(n / dt, 1), "algorithmic_GBps": round(64 * n / dt / 1e9, 1)}
    d.free()
print(json.dumps(out))
