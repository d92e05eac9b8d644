#!/usr/bin/env python3
"""Felt-vector codec on the device (csrc/wire.hip): decode throughput with the hex text already resident in HBM, against the HBM
roofline (algorithmic bytes: 64 hex characters in + 32 bytes out = 96 B per felt)."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from noir_backend_using_gnark_amd import _lib, wire  # noqa: E402

L = _lib.lib()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << log_n
src = _lib.DeviceBuffer(n * 32)
_lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(src.ptr), C.c_size_t(n), C.c_uint64(7), C.c_int(1), C.c_int(0), None))
text = wire.serialize_felts(src, n)                      # device encode -> host text
assert len(text) == 8 + 64 * n and int(text[:8], 16) == n
d_text = _lib.DeviceBuffer(len(text) + 16)
_lib.check(L.zk_dev_h2d(C.c_void_p(d_text.ptr + 8), C.c_char_p(text.encode()), C.c_size_t(len(text))))
out = _lib.DeviceBuffer(n * 32)
call = lambda: _lib.check(L.zk_bn254_felts_decode_hex_dev(C.c_void_p(d_text.ptr + 8), C.c_size_t(len(text)), C.c_void_p(out.ptr), C.c_size_t(n),
                                                         C.c_size_t(n), C.c_int(1), None))
call()
assert (out.to_numpy(np.uint64, (n, 4)) == src.to_numpy(np.uint64, (n, 4))).all(), "decode(encode(x)) != x"
_lib.profile(True)
_lib.profile_reset()
reps = 20
t0 = time.perf_counter()
for _ in range(reps):
    call()
wall = (time.perf_counter() - t0) / reps
_lib.profile(False)
k = _lib.profile_read()["felts_decode_hex"]
ker = k[1] / k[0] * 1e-3
t0 = time.perf_counter()
d2, m = wire.deserialize_felts(text)                      # host text -> HBM (PCIe-inclusive)
host_path = time.perf_counter() - t0
print(json.dumps({"workload": "felt-vector decode 2^%d" % log_n, "kernel_ms": round(ker * 1e3, 4), "felts_per_s": round(n / ker, 1),
                  "roofline": {"bound": "hbm", "achieved": round(96 * n / ker / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(96 * n / ker / 8e12, 4)},
                  "call_ms_incl_sync": round(wall * 1e3, 4), "host_text_to_hbm_ms": round(host_path * 1e3, 2), "roundtrip_exact": True}))
