#!/usr/bin/env python3
"""Stand-alone computeH (zk_bn254_groth16_compute_h_dev: three resident input vectors -> h) at 2^20 / 2^22 / 2^24: best and median wall time of 30 synchronous
calls, one JSON line.  After _lib.use_library(<csrc/build_exp/libzkmi_exp.so>) the experiment switches of ntt.hip (ZKMI_H_BATCH, ZKMI_H_FUSE_PW, ...) apply; the sha of h shows that
every variant computes the same bytes.
    python tools/compute_h_bench.py [log_n ...]"""
import ctypes as C
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from noir_backend_using_gnark_amd import _lib  # noqa: E402

L = _lib.lib()
out = {"switches": {k: v for k, v in os.environ.items() if k.startswith("ZKMI_")}}
for log_n in [int(a) for a in sys.argv[1:]] or [20, 22, 24]:
    n = 1 << log_n
    da, db, dc, dh = (_lib.DeviceBuffer(n * 32) for _ in range(4))
    for i, d in enumerate((da, db, dc)):
        _lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(d.ptr), C.c_size_t(n), C.c_uint64(10 + i), C.c_int(1), C.c_int(0), None))

    def call():
        _lib.check(L.zk_bn254_groth16_compute_h_dev(C.c_void_p(da.ptr), C.c_void_p(db.ptr), C.c_void_p(dc.ptr), C.c_size_t(n), C.c_uint32(log_n), C.c_void_p(dh.ptr), None))

    for _ in range(3):
        call()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        call()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    sha = hashlib.sha256(dh.to_numpy("uint8", (n * 32,)).tobytes()).hexdigest()[:16]
    out["2p%d" % log_n] = {"best_ms": round(ts[0], 4), "median_ms": round(ts[len(ts) // 2], 4), "h_sha": sha}
    for d in (da, db, dc, dh):
        d.free()
print(json.dumps(out))
