#!/usr/bin/env python3
"""SURVEY §8 row f1 measured for the Groth16 key: groth16.ProvingKey.WriteTo / ReadFrom (the hex payload of the reference's intended ProveWithPK,
gnark_backend_ffi/backend/groth16/r1cs.go:107-128, deserialised on EVERY call there) with every point compressed / decompressed on the device.
usage: key_bench.py [log_n=20]   -> one JSON line.  Checks (no oracle): WriteTo(ReadFrom(x)) == x and the re-read key proves the same 128 bytes."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from noir_backend_using_gnark_amd import _lib, groth16 as zk  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
L = _lib.lib()
inst = bench.Instance(L, _lib, zk, log_n, 0, 8, 0, True)
N = 1 << log_n


def prove(pk):
    return zk.prove(pk, inst.d_a, inst.d_b, inst.d_c, inst.d_w, inst.r, inst.s, n_constraints=N, on_device=True, n_wires=N)


want = prove(inst.pk)
out = {"log_n": log_n, "g1_points": 4 * N - 8, "g2_points": N}
img = {}
for name, as_hex in (("bytes", False), ("hex", True)):
    inst.pk.write_to(as_hex=as_hex)  # warm
    _lib.profile(True)
    _lib.profile_reset()
    t0 = time.perf_counter()
    img[name] = inst.pk.write_to(as_hex=as_hex)
    dt = time.perf_counter() - t0
    _lib.profile(False)
    prof = _lib.profile_read()
    out["write_" + name] = {"wall_ms": round(dt * 1e3, 1), "size_bytes": len(img[name]),
                            "kernel_ms": {k: round(v[1], 3) for k, v in prof.items() if k in ("g1_compress", "g2_compress", "hex_encode", "inf_flags")}}
assert img["hex"] == img["bytes"].hex()
for name, tables in (("bytes", False), ("hex", False), ("hex", True)):
    zk.ProvingKey.read_from(img[name], is_hex=name == "hex", precompute_tables=tables).free()  # warm
    _lib.profile(True)
    _lib.profile_reset()
    t0 = time.perf_counter()
    rk = zk.ProvingKey.read_from(img[name], is_hex=name == "hex", precompute_tables=tables)
    dt = time.perf_counter() - t0
    _lib.profile(False)
    prof = _lib.profile_read()
    assert prove(rk) == want, "the re-read key proves other bytes"
    if not tables:
        assert rk.write_to() == img["bytes"], "WriteTo(ReadFrom(x)) != x"
    rk.free()
    km = {k: round(v[1], 3) for k, v in prof.items() if k in ("g1_decompress", "g2_decompress", "hex_decode", "msm_build_table_g1", "msm_build_table_g2", "msm_expand_bases")}
    e = {"wall_ms": round(dt * 1e3, 1), "kernel_ms": km}
    if "g1_decompress" in km:
        e["g1_points_per_s"] = round((4 * N - 8 + 3) / (km["g1_decompress"] * 1e-3), 0)
        e["g2_points_per_s"] = round((N + 2) / (km["g2_decompress"] * 1e-3), 0)
    out["read_" + name + ("_with_tables" if tables else "")] = e
print(json.dumps(out))
