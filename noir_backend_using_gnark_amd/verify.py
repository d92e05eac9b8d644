"""Host-side verifiers of libzkmi (no GPU needed): groth16.Verify, plonk.Verify and the pairing check they are built on -- the counterpart of the
reference's PlonkVerifyWithVK (gnark_backend_ffi/main.go:44-56) and the intended Groth16 VerifyWithVK (backend/groth16/r1cs.go:176-212).
Proofs and keys are gnark's wire images (Proof.WriteTo / VerifyingKey.WriteTo as bytes, or hex str); public inputs are Montgomery limbs (n, 4) uint64."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib, vp


def _blob(x):
    if isinstance(x, str):
        return x.encode("ascii"), 1
    return bytes(x), 0


def _call(fn, *args) -> bool:
    ok = C.c_int(0)
    rc = fn(*args, C.byref(ok))
    if rc in (_lib.ZK_ERR_LEN, _lib.ZK_ERR_ARG):
        raise ValueError((lib().zk_last_error() or b"").decode())
    check(rc)
    return bool(ok.value)


def groth16_verify(proof: bytes, vk, public_inputs) -> bool:
    """groth16.Verify(proof, vk, publicWitness); public_inputs WITHOUT the constant wire."""
    proof = bytes(proof)
    if len(proof) != 128:
        raise ValueError("a Groth16 proof is 128 bytes (Proof.WriteTo)")
    k, is_hex = _blob(vk)
    pub = np.ascontiguousarray(public_inputs, dtype=np.uint64).reshape(-1, 4)
    return _call(lib().zk_bn254_groth16_verify, C.c_char_p(proof), C.c_char_p(k), C.c_size_t(len(k)), C.c_int(is_hex), vp(pub) if len(pub) else None, C.c_size_t(len(pub)))


def plonk_verify(proof: bytes, vk, srs_g2, public_inputs) -> bool:
    """plonk.Verify(proof, vk, publicWitness) with vk.InitKZG(srs): srs_g2 = the SRS's two G2 points ((2, 16) uint64, kzg.SRS.g2)."""
    proof = bytes(proof)
    if len(proof) != _lib.PLONK_PROOF_BYTES:
        raise ValueError("a PLONK proof is %d bytes (Proof.WriteTo)" % _lib.PLONK_PROOF_BYTES)
    k, is_hex = _blob(vk)
    g2 = np.ascontiguousarray(srs_g2, dtype=np.uint64).reshape(2, 16)
    pub = np.ascontiguousarray(public_inputs, dtype=np.uint64).reshape(-1, 4)
    return _call(lib().zk_bn254_plonk_verify, C.c_char_p(proof), C.c_char_p(k), C.c_size_t(len(k)), C.c_int(is_hex), vp(g2), vp(pub) if len(pub) else None, C.c_size_t(len(pub)))


def pairing_check(g1_points, g2_points) -> bool:
    """prod_i e(P_i, Q_i) == 1 for affine Montgomery images ((n, 8) and (n, 16) uint64)."""
    p = np.ascontiguousarray(g1_points, dtype=np.uint64).reshape(-1, 8)
    q = np.ascontiguousarray(g2_points, dtype=np.uint64).reshape(-1, 16)
    if len(p) != len(q):
        raise ValueError("as many G1 as G2 points")
    return _call(lib().zk_bn254_pairing_check, vp(p) if len(p) else None, vp(q) if len(q) else None, C.c_size_t(len(p)))
