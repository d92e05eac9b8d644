"""Host-side mirror of gnark v0.8.0's PLONK backend for the hot path (pinned at /root/reference/gnark_backend_ffi/go.mod:23):
    plonk.Setup(spr, srs)          reached at /root/reference/gnark_backend_ffi/backend/plonk/plonk.go:21   -> setup(circuit, srs)
    plonk.Prove(spr, pk, witness)  reached at backend/plonk/plonk.go:67 (PlonkProveWithPK, main.go:24-37)  -> prove(pk, solution, blinders)
The constraint system is the reference's: one gate qL*xa + qR*xb + qO*xc + qM*xa*xb + qK = 0 per ACIR arithmetic opcode
(backend/plonk/sparse_r1cs.go:44-107).  `prove` starts where gnark's prover is after `spr.Solve`: the values of all variables.
The blinding scalars (upstream: fr.SetRandom) are explicit inputs; the challenges follow upstream's SHA-256 transcript unless pinned.
Everything dispatches to libzkmi.so; nothing is computed on the host here."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib, vp
from .bn254 import ResidentBases

PROOF_BYTES = _lib.PLONK_PROOF_BYTES


class Circuit:
    """cs.SparseR1CS in the shape the reference builds it: n_public public + (n_vars - n_public) secret variables, gates as arrays.
    Coefficients: (n_constraints, 4) uint64 Montgomery images (numpy) or DeviceBuffers; wire ids: uint32 numpy arrays."""

    def __init__(self, n_public: int, n_vars: int, ql, qr, qo, qm, qk, xa, xb, xc):
        self.n_public, self.n_vars = n_public, n_vars
        self.xa, self.xb, self.xc = (np.ascontiguousarray(v, dtype=np.uint32) for v in (xa, xb, xc))
        self.n_constraints = int(self.xa.shape[0])
        self.on_device = isinstance(ql, _lib.DeviceBuffer)
        self.coeffs = [ql, qr, qo, qm, qk] if self.on_device else [np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4) for v in (ql, qr, qo, qm, qk)]


class ProvingKey:
    """plonk.ProvingKey resident in HBM (canonical selectors / permutation polynomials, their Lagrange-coset forms on the big domain,
    the per-proof workspace) plus the verifying key's digests."""

    def __init__(self, handle: int, srs: ResidentBases, vk: dict | None, n_vars: int):
        self.handle, self.srs, self.vk, self.n_vars = C.c_uint64(handle), srs, vk, n_vars

    def export(self, which: int, n: int) -> np.ndarray:
        out = np.zeros((n, 4), dtype=np.uint64)
        check(lib().zk_bn254_plonk_pk_export(self.handle, C.c_int(which), vp(out), C.c_size_t(n)))
        return out

    def write(self, as_hex: bool = False) -> bytes:
        """plonk.ProvingKey.WriteTo (as_hex: the text SerializeProvingKey hands to Rust, internal/backend/helpers.go:82-87)"""
        n = C.c_size_t(0)
        lib().zk_bn254_plonk_pk_write(self.handle, C.c_int(int(as_hex)), C.create_string_buffer(1), C.c_size_t(0), C.byref(n))  # size query
        buf = C.create_string_buffer(n.value)
        check(lib().zk_bn254_plonk_pk_write(self.handle, C.c_int(int(as_hex)), buf, C.c_size_t(n.value), C.byref(n)))
        return buf.raw[:n.value]

    def lagrange_srs(self) -> None:
        """Build the SRS's Lagrange form over this key's domain (zk_bn254_plonk_pk_lagrange_srs): later proofs commit l, r, o from the wire values."""
        check(lib().zk_bn254_plonk_pk_lagrange_srs(self.handle))

    def free(self):
        if self.handle.value:
            check(lib().zk_bn254_plonk_pk_free(self.handle))
            self.handle = C.c_uint64(0)


def _vk_dict(v: _lib.PlonkVK) -> dict:
    a = lambda x: np.array(list(x), dtype=np.uint64)
    return dict(size=int(v.size), n_public=int(v.n_public), size_inv=a(v.size_inv), generator=a(v.generator), coset_shift=a(v.coset_shift),
                s=a(v.s).reshape(3, 8), ql=a(v.ql), qr=a(v.qr), qm=a(v.qm), qo=a(v.qo), qk=a(v.qk))


def setup(circuit: Circuit, srs: ResidentBases) -> ProvingKey:
    """plonk.Setup: selectors and permutation polynomials in canonical form, their commitments (the verifying key), the cached
    Lagrange-coset forms.  `srs` = kzg SRS.G1 registered with ResidentBases (>= domain size + 3 points)."""
    ptr = (lambda b: b.ptr) if circuit.on_device else (lambda a: a.ctypes.data)
    c = _lib.PlonkCircuit(circuit.n_public, circuit.n_constraints, circuit.n_vars, *[ptr(v) for v in circuit.coeffs],
                          circuit.xa.ctypes.data, circuit.xb.ctypes.data, circuit.xc.ctypes.data, 1 if circuit.on_device else 0, 0)
    h, vk = C.c_uint64(0), _lib.PlonkVK()
    check(lib().zk_bn254_plonk_setup(C.byref(c), srs.handle, C.byref(h), C.byref(vk)))
    return ProvingKey(h.value, srs, _vk_dict(vk), circuit.n_vars)


def load_proving_key(log_n: int, n_public: int, n_vars: int, polys: dict, permutation, xa, xb, xc, vk: dict, srs: ResidentBases) -> ProvingKey:
    """gnark's own ProvingKey fields (as plonk.Setup / ReadFrom leave them): polys = canonical ql, qr, qm, qo, cqk, s1, s2, s3 and lqk."""
    keep = {k: np.ascontiguousarray(polys[k], dtype=np.uint64) for k in ("ql", "qr", "qm", "qo", "cqk", "lqk", "s1", "s2", "s3")}
    perm = np.ascontiguousarray(permutation, dtype=np.int64)
    w = [np.ascontiguousarray(v, dtype=np.uint32) for v in (xa, xb, xc)]
    vs = np.ascontiguousarray(vk["s"], dtype=np.uint64)
    vq = {k: np.ascontiguousarray(vk[k], dtype=np.uint64) for k in ("ql", "qr", "qm", "qo", "qk")}
    k = _lib.PlonkPK(log_n, n_public, int(w[0].shape[0]), n_vars, *[keep[x].ctypes.data for x in ("ql", "qr", "qm", "qo", "cqk", "lqk", "s1", "s2", "s3")],
                     perm.ctypes.data, w[0].ctypes.data, w[1].ctypes.data, w[2].ctypes.data, vs.ctypes.data, *[vq[x].ctypes.data for x in ("ql", "qr", "qm", "qo", "qk")])
    h = C.c_uint64(0)
    check(lib().zk_bn254_plonk_pk_load(C.byref(k), srs.handle, C.byref(h)))
    return ProvingKey(h.value, srs, vk, n_vars)


def read_proving_key(data, n_vars: int, xa, xb, xc, srs, is_hex: bool = False) -> ProvingKey:
    """plonk.ProvingKey.ReadFrom on gnark's bytes (or the hex text DeserializeProvingKey receives, helpers.go:49-60); the wire ids are the
    rebuilt spr's (plonk.go:54).  The vectors are decoded on the device straight into the resident key."""
    raw = data.encode("ascii") if isinstance(data, str) else bytes(data)
    w = [np.ascontiguousarray(v, dtype=np.uint32) for v in (xa, xb, xc)]
    h = C.c_uint64(0)
    rc = lib().zk_bn254_plonk_pk_read(C.c_char_p(raw), C.c_size_t(len(raw)), C.c_int(int(is_hex)), C.c_size_t(n_vars), C.c_size_t(int(w[0].shape[0])),
                                      w[0].ctypes.data_as(C.c_void_p), w[1].ctypes.data_as(C.c_void_p), w[2].ctypes.data_as(C.c_void_p), srs.handle, C.byref(h))
    if rc in (_lib.ZK_ERR_LEN, _lib.ZK_ERR_ARG):
        raise ValueError((lib().zk_last_error() or b"").decode())
    check(rc)
    return ProvingKey(h.value, srs, None, n_vars)


def prove(pk: ProvingKey, solution, blinders, challenges=None) -> bytes:
    """plonk.Prove after the solver -> Proof.WriteTo bytes (548).  solution: (n_vars, 4) Montgomery values of all variables (numpy) or a
    DeviceBuffer; blinders: (9, 4); challenges: None (Fiat-Shamir as upstream) or (5, 4) = gamma, beta, alpha, zeta, kzg gamma."""
    on_dev = isinstance(solution, _lib.DeviceBuffer)
    if on_dev:
        ptr, n = C.c_void_p(solution.ptr), pk.n_vars
    else:
        sol = np.ascontiguousarray(solution, dtype=np.uint64).reshape(-1, 4)
        ptr, n = vp(sol), sol.shape[0]
    bl = np.ascontiguousarray(blinders, dtype=np.uint64).reshape(9, 4)
    ch = None if challenges is None else np.ascontiguousarray(challenges, dtype=np.uint64).reshape(5, 4)
    out = (C.c_uint8 * PROOF_BYTES)()
    rc = lib().zk_bn254_plonk_prove(pk.handle, ptr, C.c_size_t(n), C.c_int(int(on_dev)), vp(bl), vp(ch) if ch is not None else None, out)
    if rc == _lib.ZK_ERR_LEN:
        raise ValueError((lib().zk_last_error() or b"").decode())
    check(rc)
    return bytes(out)
