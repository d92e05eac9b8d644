"""Shared helpers for the parity tests (numpy <-> python ints <-> gnark memory images)."""
import hashlib

import numpy as np

from oracle import bn254_ref as ref
from oracle import oracle as orc


def h2i(h: str) -> int:
    return int(h, 16)


def mont_limbs(xs, m=ref.R) -> np.ndarray:
    """canonical python ints -> (n,4) uint64 Montgomery limbs, computed with python ints (independent of C code)."""
    out = np.zeros((len(xs), 4), dtype=np.uint64)
    for i, x in enumerate(xs):
        v = ref.to_mont(x % m, m)
        for k in range(4):
            out[i, k] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


def from_mont_limbs(a, m=ref.R):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return [ref.from_mont(sum(int(a[i, k]) << (64 * k) for k in range(4)), m) for i in range(a.shape[0])]


def sha_image(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def hex_to_u64(h: str) -> np.ndarray:
    return np.frombuffer(bytes.fromhex(h), dtype=np.uint64).copy()


def g1_points_from_scalars(ks) -> np.ndarray:
    out = np.zeros((len(ks), 8), dtype=np.uint64)
    for i, k in enumerate(ks):
        P = ref.g1_mul(ref.G1_GEN, k) if k % ref.R else None
        out[i] = np.frombuffer(ref.g1_affine_mont_bytes(P), dtype=np.uint64)
    return out


def g2_points_from_scalars(ks) -> np.ndarray:
    out = np.zeros((len(ks), 16), dtype=np.uint64)
    for i, k in enumerate(ks):
        P = ref.g2_mul(ref.G2_GEN, k) if k % ref.R else None
        out[i] = np.frombuffer(ref.g2_affine_mont_bytes(P), dtype=np.uint64)
    return out


def golden_pk(entry) -> dict:
    pk = {k: hex_to_u64(v) for k, v in entry["pk"].items() if k != "log_domain"}
    pk["log_domain"] = entry["pk"]["log_domain"]
    pk["n_wires"] = entry["n_wires"]
    pk["n_public"] = entry["n_public"]
    return pk
