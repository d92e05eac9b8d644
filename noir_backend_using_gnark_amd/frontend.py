"""Host-side mirror of the reference's exported Go entry points (gnark_backend_ffi/main.go:24-78), restated over the device path:
    PlonkPreprocess(acirJSON, encodedValues) -> (hex pk, hex vk)          -> plonk_preprocess
    PlonkProveWithPK(acirJSON, encodedValues, encodedProvingKey) -> hex   -> plonk_prove_with_pk
    BuildSparseR1CS (backend/plonk/sparse_r1cs.go:18-107)                 -> acir_to_sparse_r1cs
and of the intended Groth16 FFI (backend/groth16/r1cs.go:74-266, commented out upstream):
    Preprocess(rawR1CS) -> (hex pk, hex vk)                               -> groth16_preprocess
    ProveWithPK(rawR1CS, encodedProvingKey) -> hex proof                  -> groth16_prove_with_pk
    ProveWithMeta(rawR1CS) -> hex proof                                   -> groth16_prove_with_meta
Strings in, strings out, like the Rust side sees them (src/gnark_backend_wrapper/plonk/mod.rs:19-23, 207); the SRS is a resident
`kzg.SRS` instead of the srs.hex file the reference re-reads on every call.  Everything dispatches to libzkmi.so."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib, vp


# how witnesses become variables (include/zkmi.h): HandleValues as written (backend/common.go:45-76) / one variable per witness
LAYOUT_REFERENCE, LAYOUT_ONE_VAR_PER_WITNESS = 0, 1


def _b(s) -> bytes:
    return s.encode("ascii") if isinstance(s, str) else bytes(s)


def plonk_preprocess(acir_json: str, encoded_values: str, srs, keep_resident: bool = False, layout: int = LAYOUT_REFERENCE):
    """-> (pk_hex, vk_hex[, resident ProvingKey handle])"""
    a, v = _b(acir_json), _b(encoded_values)
    pk_len, vk_len = C.c_size_t(0), C.c_size_t(0)
    check(lib().zk_plonk_preprocess(C.c_char_p(a), C.c_size_t(len(a)), C.c_char_p(v), C.c_size_t(len(v)), C.c_int(layout), srs.handle, None, C.c_size_t(0), C.byref(pk_len),
                                    None, C.c_size_t(0), C.byref(vk_len), None))
    pk, vk = C.create_string_buffer(pk_len.value), C.create_string_buffer(vk_len.value)
    h = C.c_uint64(0)
    check(lib().zk_plonk_preprocess(C.c_char_p(a), C.c_size_t(len(a)), C.c_char_p(v), C.c_size_t(len(v)), C.c_int(layout), srs.handle, pk, C.c_size_t(pk_len.value), C.byref(pk_len),
                                    vk, C.c_size_t(vk_len.value), C.byref(vk_len), C.byref(h) if keep_resident else None))
    out = (pk.raw[:pk_len.value].decode(), vk.raw[:vk_len.value].decode())
    return out + (h.value,) if keep_resident else out


def plonk_prove_with_pk(acir_json: str, encoded_values: str, encoded_pk: str | None, srs, blinders=None, pk_handle: int = 0, layout: int = LAYOUT_REFERENCE) -> str:
    """-> hex of Proof.WriteTo.  encoded_pk=None proves with the resident key `pk_handle`; blinders=None draws them like upstream."""
    a, v = _b(acir_json), _b(encoded_values)
    k = _b(encoded_pk) if encoded_pk is not None else None
    bl = None if blinders is None else np.ascontiguousarray(blinders, dtype=np.uint64).reshape(9, 4)
    out = C.create_string_buffer(2 * _lib.PLONK_PROOF_BYTES)
    rc = lib().zk_plonk_prove_with_pk(C.c_char_p(a), C.c_size_t(len(a)), C.c_char_p(v), C.c_size_t(len(v)), C.c_int(layout), C.c_char_p(k) if k is not None else None,
                                      C.c_size_t(len(k) if k is not None else 0), C.c_uint64(pk_handle), srs.handle, vp(bl) if bl is not None else None, out)
    if rc in (_lib.ZK_ERR_LEN, _lib.ZK_ERR_ARG):
        raise ValueError((lib().zk_last_error() or b"").decode())
    check(rc)
    return out.raw.decode()


def acir_to_sparse_r1cs(acir_json: str, n_values: int, layout: int = LAYOUT_REFERENCE) -> dict:
    """BuildSparseR1CS + HandleValues: gates (Montgomery coefficients, variable ids) and the witness order (variable k = witness order[k] + 1)."""
    a = _b(acir_json)
    npub, nvars, nc = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    rc = lib().zk_acir_to_sparse_r1cs(C.c_char_p(a), C.c_size_t(len(a)), C.c_size_t(n_values), C.c_int(layout), C.byref(npub), C.byref(nvars), C.byref(nc), *([None] * 9))
    if rc == _lib.ZK_ERR_ARG:
        raise ValueError((lib().zk_last_error() or b"").decode())
    check(rc)
    co = [np.zeros((nc.value, 4), np.uint64) for _ in range(5)]
    wi = [np.zeros(nc.value, np.uint32) for _ in range(3)]
    order = np.zeros(nvars.value, np.uint32)
    check(lib().zk_acir_to_sparse_r1cs(C.c_char_p(a), C.c_size_t(len(a)), C.c_size_t(n_values), C.c_int(layout), None, None, None, *[vp(x) for x in co], *[vp(x) for x in wi], vp(order)))
    return dict(n_public=npub.value, n_vars=nvars.value, ql=co[0], qr=co[1], qo=co[2], qm=co[3], qk=co[4], xa=wi[0], xb=wi[1], xc=wi[2], order=order)


def groth16_r1cs_from_raw(raw_json: str):
    """buildR1CS on the reference's RawR1CS JSON (backend/groth16/r1cs.go:9-72, src/gnark_backend_wrapper/groth16/acir_to_r1cs.rs:18-60)
    -> (groth16.R1CS resident in HBM, DeviceBuffer holding the full wire vector [ONE, public, secret, product variables])."""
    from .groth16 import R1CS
    a = _b(raw_json)
    h, d, nw, npub = C.c_uint64(0), C.c_void_p(0), C.c_size_t(0), C.c_size_t(0)
    rc = lib().zk_groth16_r1cs_from_raw(C.c_char_p(a), C.c_size_t(len(a)), C.byref(h), C.byref(d), C.byref(nw), C.byref(npub))
    if rc in (_lib.ZK_ERR_LEN, _lib.ZK_ERR_ARG):
        raise ValueError((lib().zk_last_error() or b"").decode())
    check(rc)
    buf = _lib.DeviceBuffer.__new__(_lib.DeviceBuffer)
    buf.ptr, buf.nbytes = int(d.value), nw.value * 32
    return R1CS.from_handle(h.value, npub.value, nw.value, -1), buf


def _fr_arr(x, n):
    return None if x is None else np.ascontiguousarray(x, dtype=np.uint64).reshape(n, 4)


def groth16_preprocess(raw_json: str, toxic=None, keep_resident: bool = False):
    """Preprocess (r1cs.go:214-266): groth16.Setup on the RawR1CS -> (pk_hex, vk_hex[, resident key handle]).  toxic: (5, 4) Montgomery tau, alpha,
    beta, gamma, delta; None draws them like upstream (then the sizing call and the real one build different keys of the same size)."""
    a = _b(raw_json)
    tx = _fr_arr(toxic, 5)
    pk_len, vk_len = C.c_size_t(0), C.c_size_t(0)
    rc = lib().zk_groth16_preprocess(C.c_char_p(a), C.c_size_t(len(a)), vp(tx) if tx is not None else None, None, C.c_size_t(0), C.byref(pk_len), None, C.c_size_t(0),
                                     C.byref(vk_len), None)
    if rc in (_lib.ZK_ERR_LEN, _lib.ZK_ERR_ARG):
        raise ValueError((lib().zk_last_error() or b"").decode())
    check(rc)
    pk, vk = C.create_string_buffer(pk_len.value), C.create_string_buffer(vk_len.value)
    h = C.c_uint64(0)
    check(lib().zk_groth16_preprocess(C.c_char_p(a), C.c_size_t(len(a)), vp(tx) if tx is not None else None, pk, C.c_size_t(pk_len.value), C.byref(pk_len), vk,
                                      C.c_size_t(vk_len.value), C.byref(vk_len), C.byref(h) if keep_resident else None))
    out = (pk.raw[:pk_len.value].decode(), vk.raw[:vk_len.value].decode())
    return out + (h.value,) if keep_resident else out


def groth16_prove_with_pk(raw_json: str, encoded_pk: str | None, rs=None, pk_handle: int = 0) -> str:
    """ProveWithPK (r1cs.go:107-143) -> hex of Proof.WriteTo (256 characters).  encoded_pk=None proves with the resident key `pk_handle`."""
    a = _b(raw_json)
    k = _b(encoded_pk) if encoded_pk is not None else None
    r2 = _fr_arr(rs, 2)
    out = C.create_string_buffer(256)
    rc = lib().zk_groth16_prove_with_pk(C.c_char_p(a), C.c_size_t(len(a)), C.c_char_p(k) if k is not None else None, C.c_size_t(len(k) if k is not None else 0),
                                        C.c_uint64(pk_handle), vp(r2) if r2 is not None else None, out)
    if rc in (_lib.ZK_ERR_LEN, _lib.ZK_ERR_ARG):
        raise ValueError((lib().zk_last_error() or b"").decode())
    check(rc)
    return out.raw.decode()


def groth16_prove_with_meta(raw_json: str, toxic=None, rs=None) -> str:
    """ProveWithMeta (r1cs.go:74-105): Setup + Prove in one call -> hex of Proof.WriteTo."""
    a = _b(raw_json)
    tx, r2 = _fr_arr(toxic, 5), _fr_arr(rs, 2)
    out = C.create_string_buffer(256)
    rc = lib().zk_groth16_prove_with_meta(C.c_char_p(a), C.c_size_t(len(a)), vp(tx) if tx is not None else None, vp(r2) if r2 is not None else None, out)
    if rc in (_lib.ZK_ERR_LEN, _lib.ZK_ERR_ARG):
        raise ValueError((lib().zk_last_error() or b"").decode())
    check(rc)
    return out.raw.decode()


def groth16_lower_resident(raw_json: str, to_device: bool = False) -> None:
    """Reads a RawR1CS text into the resident cache (host only unless to_device): the zk_groth16_* call that follows finds the circuit by content."""
    a = _b(raw_json)
    rc = lib().zk_groth16_lower_resident(C.c_char_p(a), C.c_size_t(len(a)), C.c_int(int(to_device)))
    if rc in (_lib.ZK_ERR_LEN, _lib.ZK_ERR_ARG):
        raise ValueError((lib().zk_last_error() or b"").decode())
    check(rc)


def groth16_key_resident(encoded_pk: str) -> None:
    """Decodes a key text into the resident cache (no window tables: they come with the key's second proof)."""
    k = _b(encoded_pk)
    rc = lib().zk_groth16_key_resident(C.c_char_p(k), C.c_size_t(len(k)))
    if rc in (_lib.ZK_ERR_LEN, _lib.ZK_ERR_ARG):
        raise ValueError((lib().zk_last_error() or b"").decode())
    check(rc)


def groth16_public_inputs(raw_json: str) -> np.ndarray:
    """buildWitnesses' public part for VerifyWithVK (r1cs.go:176-212): the values of the public wires after ONE, in wire order -> (n, 4) Montgomery limbs."""
    a = _b(raw_json)
    n = C.c_size_t(0)
    out = np.zeros((16, 4), dtype=np.uint64)
    rc = lib().zk_groth16_public_inputs(C.c_char_p(a), C.c_size_t(len(a)), vp(out), C.c_size_t(16), C.byref(n))
    if rc == _lib.ZK_ERR_ARG and n.value > 16:
        out = np.zeros((n.value, 4), dtype=np.uint64)
        rc = lib().zk_groth16_public_inputs(C.c_char_p(a), C.c_size_t(len(a)), vp(out), C.c_size_t(n.value), C.byref(n))
    if rc in (_lib.ZK_ERR_LEN, _lib.ZK_ERR_ARG):
        raise ValueError((lib().zk_last_error() or b"").decode())
    check(rc)
    return out[:n.value]


def export_cache_info() -> dict:
    """What the export path keeps resident between calls (PLONK and Groth16 entries together)."""
    nc, nk, by = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    check(lib().zk_export_cache_info(C.byref(nc), C.byref(nk), C.byref(by)))
    return {"circuits": nc.value, "keys": nk.value, "bytes": by.value}


def export_cache_clear() -> None:
    check(lib().zk_export_cache_clear())
