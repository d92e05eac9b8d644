#!/usr/bin/env python3
"""SURVEY §8 row f1 measured: loading a serialised KZG SRS (kzg.SRS.ReadFrom; the reference re-reads srs.hex -- 10^6 points -- on every prove /
verify call, gnark_backend_ffi/backend/plonk/plonk.go:16,34,58 -> backend/common.go:86-105) with the G1 points decompressed on the device.
usage: srs_bench.py [n_points=1000000]   -> one JSON line (no oracle involved: the round trip WriteTo(ReadFrom(x)) == x is the check)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from noir_backend_using_gnark_amd import _lib, kzg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
alpha = np.array([0x1234567, 0x89abcdef, 0x1111, 0x0222], dtype=np.uint64)  # any Montgomery image < r
t0 = time.perf_counter()
srs = kzg.new_srs(n, alpha, table_window_bits=-1)
t_new = time.perf_counter() - t0
raw = srs.write()
hexed = srs.write(as_hex=True)
srs.free()
out = {"points": n, "bytes": len(raw), "new_srs_s": round(t_new, 3)}
for name, data, is_hex in (("bytes", raw, False), ("hex", hexed, True)):
    kzg.read_srs(data, is_hex=is_hex, table_window_bits=-1).free()  # warm
    _lib.profile(True)
    _lib.profile_reset()
    t0 = time.perf_counter()
    s2 = kzg.read_srs(data, is_hex=is_hex, table_window_bits=-1)
    dt = time.perf_counter() - t0
    _lib.profile(False)
    prof = _lib.profile_read()
    assert s2.write() == raw, "WriteTo(ReadFrom(x)) != x"
    s2.free()
    out["read_" + name] = {"wall_ms": round(dt * 1e3, 2), "points_per_s": round(n / dt, 1),
                           "kernel_ms": {k: round(v[1], 3) for k, v in prof.items() if k in ("g1_decompress", "hex_decode")},
                           "decompress_algorithmic_GBps": round(96.0 * n / (prof["g1_decompress"][1] * 1e-3) / 1e9, 1)}
# with the window tables the prover uses (built once per load)
t0 = time.perf_counter()
s3 = kzg.read_srs(raw)
out["read_bytes_with_window_tables_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
s3.free()
print(json.dumps(out))
