"""Felt-vector wire codec, host-side mirror of the reference's helpers
(`DeserializeFelts` in gnark_backend_ffi/internal/backend/helpers.go:24-33, `encode_felts` in
src/gnark_backend_wrapper/serialize.rs:33-47): hex(u32 BE count || count x 32 B BE canonical felts) <-> a Montgomery
fr.Element vector resident in HBM (decoded on the device by one fused kernel, csrc/wire.hip)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib


def deserialize_felts(encoded: str | bytes, capacity: int | None = None) -> tuple[_lib.DeviceBuffer, int]:
    """DeserializeFelts: returns (device buffer of n Montgomery fr.Elements, n).  Raises on bad hex, a length that does not match
    the count, or a non-canonical felt (gnark-crypto: "invalid fr.Element encoding")."""
    text = encoded.encode("ascii") if isinstance(encoded, str) else bytes(encoded)
    if len(text) < 8:
        raise ValueError("felt vector shorter than its 4-byte count")
    cap = capacity if capacity is not None else max(1, (len(text) - 8) // 64)
    out = _lib.DeviceBuffer(cap * 32)
    n = C.c_size_t(0)
    check(lib().zk_bn254_felts_decode_hex(C.c_char_p(text), C.c_size_t(len(text)), C.c_void_p(out.ptr), C.c_size_t(cap), C.byref(n)))
    return out, int(n.value)


def serialize_felts(d_vec, n: int) -> str:
    """encode_felts of a Montgomery vector resident in HBM (DeviceBuffer or raw device pointer)."""
    ptr = d_vec.ptr if isinstance(d_vec, _lib.DeviceBuffer) else int(d_vec)
    buf = C.create_string_buffer(8 + 64 * n)
    check(lib().zk_bn254_felts_encode_hex(C.c_void_p(ptr), C.c_size_t(n), buf, C.c_size_t(8 + 64 * n)))
    return buf.raw.decode("ascii")


def deserialize_felts_to_numpy(encoded) -> np.ndarray:
    d, n = deserialize_felts(encoded)
    return d.to_numpy(np.uint64, (max(n, 1), 4))[:n]
