"""Host-side mirror of gnark v0.8.0 groth16 prove for the hot path (pinned at /root/reference/gnark_backend_ffi/go.mod:23;
the reference's live call is groth16.Prove at /root/reference/gnark_backend_ffi/main.go:131; the FFI it intended is the
commented-out ProveWithPK at backend/groth16/r1cs.go:107-143).  `prove` takes what gnark's prover has after
`r1cs.Solve`: the evaluation vectors a, b, c and the wire values w -- plus the randomness (r, s) as explicit inputs."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import Groth16PK, check, lib, vp


class ProvingKey:
    """groth16.ProvingKey{G1{Alpha,Beta,Delta,A,B,K,Z}, G2{Beta,Delta,B}} resident in HBM.

    Arrays are gnark memory images (numpy uint64) or, with bases_on_device=True, raw device pointers for the five
    base arrays (used by the benchmark, whose bases are generated on the device).

    gnark's own layout: pass infinity_a / infinity_b (the []bool InfinityA / InfinityB, one entry per wire); g1_a then holds the
    n_wires - NbInfinityA points gnark stores, g1_b / g2_b the n_wires - NbInfinityB ones (setup.go).  Without the bitmaps the
    arrays are wire-indexed with (0,0) for a point at infinity.

    window_shard=(rank, count): the WINDOW-sharded key of one rank -- the whole key, but only the table rows (digit windows) w = rank + k * count;
    msm5 against it yields the partial sums over those windows of the full wire / h vectors.

    device_mask: several GPUs in ONE process (zk_init_devices): bit i = device entry i holds a range slice of the key (2, 4 or 8 entries);
    prove() on such a key runs computeH block-sharded over the entries and the five MSMs per slice -- same bytes.

    A rank of a range-sharded proof loads ITS slice as a key of its own (n_wires / n_public / log_domain describe the
    slice; shard_full_z=True on every rank but the last, whose Z slice ends with the unused N-th entry)."""

    def __init__(self, log_domain: int, n_wires: int, n_public: int, g1_alpha, g1_beta, g1_delta, g1_a, g1_b, g1_k, g1_z,
                 g2_beta, g2_delta, g2_b, bases_on_device: bool = False, precompute_tables: bool = True, shard_full_z: bool = False,
                 infinity_a=None, infinity_b=None, table_window_bits: int = 0, window_shard: tuple | None = None, device_mask: int = 0):
        self.log_domain, self.n_wires, self.n_public = log_domain, n_wires, n_public
        self._keep = []
        inf_a = inf_b = 0
        nb_a = nb_b = 0
        if (infinity_a is None) != (infinity_b is None):
            raise ValueError("InfinityA and InfinityB come together")
        if infinity_a is not None:
            ia = np.ascontiguousarray(np.asarray(infinity_a).astype(np.uint8))
            ib = np.ascontiguousarray(np.asarray(infinity_b).astype(np.uint8))
            if ia.size != n_wires or ib.size != n_wires:
                raise ValueError("InfinityA / InfinityB need one entry per wire")
            self._keep += [ia, ib]
            inf_a, inf_b = ia.ctypes.data, ib.ctypes.data
            nb_a, nb_b = int(np.count_nonzero(ia)), int(np.count_nonzero(ib))

        def host(a):
            a = np.ascontiguousarray(a, dtype=np.uint64)
            self._keep.append(a)
            return a.ctypes.data

        def base(a):
            if bases_on_device:
                return int(a.ptr if isinstance(a, _lib.DeviceBuffer) else a)
            return host(a) if np.asarray(a).size else 0

        pk = Groth16PK(log_domain, n_wires, n_public, host(g1_alpha), host(g1_beta), host(g1_delta), base(g1_a), base(g1_b),
                       base(g1_k), base(g1_z), host(g2_beta), host(g2_delta), base(g2_b), 1 if bases_on_device else 0,
                       (0 if precompute_tables else 1) | (2 if shard_full_z else 0) | (4 if window_shard else 0), inf_a, inf_b, nb_a, nb_b, table_window_bits, device_mask,
                       *(window_shard or (0, 0)))
        if bases_on_device:
            self._keep += [g1_a, g1_b, g1_k, g1_z, g2_b]  # keep DeviceBuffers alive
        self.handle = C.c_uint64(0)
        check(lib().zk_bn254_groth16_pk_load(C.byref(pk), C.byref(self.handle)))

    @classmethod
    def from_handle(cls, handle: int, log_domain: int, n_wires: int, n_public: int):
        """wraps a key that already lives in the library (groth16.Setup on the device)"""
        pk = cls.__new__(cls)
        pk.log_domain, pk.n_wires, pk.n_public, pk._keep, pk.handle = log_domain, n_wires, n_public, [], C.c_uint64(handle)
        return pk

    @classmethod
    def read_from(cls, data, is_hex: bool | None = None, precompute_tables: bool = True, table_window_bits: int = 0):
        """groth16.ProvingKey.ReadFrom on the bytes of ProvingKey.WriteTo (or their hex text, the form the reference's ProveWithPK receives:
        backend/groth16/r1cs.go:107-128): points are decompressed on the device."""
        raw = data.encode("ascii") if isinstance(data, str) else bytes(data)
        if is_hex is None:
            is_hex = isinstance(data, str)
        h = C.c_uint64(0)
        rc = lib().zk_bn254_groth16_pk_read(C.c_char_p(raw), C.c_size_t(len(raw)), C.c_int(int(is_hex)), C.c_int(0 if precompute_tables else 1),
                                            C.c_int(table_window_bits), C.byref(h))
        if rc in (_lib.ZK_ERR_LEN, _lib.ZK_ERR_ARG):
            raise ValueError((lib().zk_last_error() or b"").decode())
        check(rc)
        pk = cls.from_handle(h.value, 0, 0, 0)
        i = pk.info()
        pk.log_domain, pk.n_wires, pk.n_public = i["log_domain"], i["n_wires"], i["n_public"]
        return pk

    def write_to(self, as_hex: bool = False):
        """groth16.ProvingKey.WriteTo (compressed points; A / B / G2.B without their points at infinity + InfinityA / InfinityB) -> bytes, or hex str"""
        n = C.c_size_t(0)
        check(lib().zk_bn254_groth16_pk_write(self.handle, C.c_int(int(as_hex)), None, C.c_size_t(0), C.byref(n)))
        buf = C.create_string_buffer(n.value)
        check(lib().zk_bn254_groth16_pk_write(self.handle, C.c_int(int(as_hex)), buf, C.c_size_t(n.value), C.byref(n)))
        return buf.raw[:n.value].decode() if as_hex else buf.raw[:n.value]

    def vk_write_to(self, vk: dict, as_hex: bool = False):
        """groth16.VerifyingKey.WriteTo from setup()'s vk dict (this key supplies [beta]1 and [delta]1)"""
        g1 = np.ascontiguousarray(np.concatenate([np.asarray(vk["g1_alpha"], np.uint64).reshape(1, 8), np.asarray(vk["g1_k"], np.uint64).reshape(-1, 8)]))
        g2 = np.ascontiguousarray(np.stack([np.asarray(vk[k], np.uint64).reshape(16) for k in ("g2_beta", "g2_gamma", "g2_delta")]))
        n = C.c_size_t(0)
        cap = 2 * (292 + 32 * (g1.shape[0] - 1))
        buf = C.create_string_buffer(cap)
        rc = lib().zk_bn254_groth16_vk_write(self.handle, vp(g1), C.c_size_t(g1.shape[0] - 1), vp(g2), C.c_int(int(as_hex)), buf, C.c_size_t(cap), C.byref(n))
        if rc == _lib.ZK_ERR_LEN:
            raise ValueError((lib().zk_last_error() or b"").decode())
        check(rc)
        return buf.raw[:n.value].decode() if as_hex else buf.raw[:n.value]

    def info(self) -> dict:
        nw, npub, ld, tab = C.c_size_t(0), C.c_size_t(0), C.c_uint32(0), C.c_int(0)
        check(lib().zk_bn254_groth16_pk_info(self.handle, C.byref(nw), C.byref(npub), C.byref(ld), C.byref(tab)))
        return dict(n_wires=int(nw.value), n_public=int(npub.value), log_domain=int(ld.value), tables=bool(tab.value))

    def free(self):
        if self.handle.value:
            check(lib().zk_bn254_groth16_pk_free(self.handle))
            self.handle = C.c_uint64(0)


def _ptr(x):
    if isinstance(x, _lib.DeviceBuffer):
        return C.c_void_p(x.ptr)
    if isinstance(x, int):
        return C.c_void_p(x)
    return vp(np.ascontiguousarray(x, dtype=np.uint64))


def prove(pk: ProvingKey, a, b, c, w, r, s, n_constraints: int | None = None, on_device: bool = False, n_wires: int | None = None) -> bytes:
    """groth16.Prove from the solver output: returns Proof.WriteTo bytes (Ar | Bs | Krs compressed, 128 B).
    len(w) != the key's wire count is the library's error (ZK_ERR_LEN -> ValueError), like a mismatched MultiExp upstream."""
    if not on_device:
        a, b, c, w = (np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4) for v in (a, b, c, w))
        if not (a.shape == b.shape == c.shape):
            raise ValueError("a, b, c must have the same length")
        n_constraints = a.shape[0]
        n_wires = w.shape[0]
    elif n_wires is None:
        n_wires = pk.n_wires  # device pointers carry no length: the caller vouches for n_wires elements
    r = np.ascontiguousarray(r, dtype=np.uint64).reshape(4)
    s = np.ascontiguousarray(s, dtype=np.uint64).reshape(4)
    proof = (C.c_uint8 * 128)()
    rc = lib().zk_bn254_groth16_prove(pk.handle, _ptr(a), _ptr(b), _ptr(c), C.c_size_t(n_constraints), _ptr(w), C.c_size_t(n_wires), vp(r), vp(s),
                                      C.c_int(int(on_device)), proof)
    if rc == _lib.ZK_ERR_LEN:
        raise ValueError((lib().zk_last_error() or b"").decode())
    check(rc)
    return bytes(proof)


def compute_h(a, b, c, log_domain: int) -> np.ndarray:
    """gnark's computeH(a, b, c, domain): (N, 4) coefficients of h in the (bit-reversed) order upstream leaves them."""
    a, b, c = (np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4) for v in (a, b, c))
    if not (a.shape == b.shape == c.shape):
        raise ValueError("a, b, c must have the same length")
    h = np.zeros((1 << log_domain, 4), dtype=np.uint64)
    check(lib().zk_bn254_groth16_compute_h(vp(a), vp(b), vp(c), C.c_size_t(a.shape[0]), C.c_uint32(log_domain), vp(h)))
    return h


class R1CS:
    """cs.R1CS resident in HBM: constraints (L w) o (R w) = (O w) over the wires [ONE, public..., secret..., internal...]; n_public counts the
    ONE wire (gnark's GetNbPublicVariables).  `constraints`: list of (L, R, O), each a dict wire -> Montgomery coefficient (4 limbs)."""

    def __init__(self, n_public: int, n_wires: int, constraints):
        self.n_public, self.n_wires, self.n_constraints = n_public, n_wires, len(constraints)
        mats = []
        for m in range(3):
            ptr, idx, val = [0], [], []
            for con in constraints:
                for wire, coeff in con[m].items():
                    idx.append(wire)
                    val.append(np.asarray(coeff, dtype=np.uint64).reshape(4))
                ptr.append(len(idx))
            mats.append((np.asarray(ptr, dtype=np.uint32), np.asarray(idx, dtype=np.uint32), np.ascontiguousarray(np.stack(val)) if val else np.zeros((0, 4), np.uint64)))
        self._keep = mats
        raw = _lib.R1CS(self.n_constraints, n_wires, n_public, *[a.ctypes.data for mat in mats for a in mat])
        self.handle = C.c_uint64(0)
        check(lib().zk_bn254_r1cs_load(C.byref(raw), C.byref(self.handle)))

    @classmethod
    def from_handle(cls, handle: int, n_public: int, n_wires: int, n_constraints: int):
        r = cls.__new__(cls)
        r.n_public, r.n_wires, r.n_constraints, r._keep, r.handle = n_public, n_wires, n_constraints, [], C.c_uint64(handle)
        return r

    def eval_abc(self, w) -> tuple:
        """a, b, c = L w, R w, O w on the device (the solver's output for a system without hints) -> numpy"""
        w = np.ascontiguousarray(w, dtype=np.uint64).reshape(-1, 4)
        dw = _lib.DeviceBuffer.from_numpy(w)
        out = [_lib.DeviceBuffer(max(self.n_constraints, 1) * 32) for _ in range(3)]
        rc = lib().zk_bn254_r1cs_eval_abc_dev(self.handle, C.c_void_p(dw.ptr), C.c_size_t(w.shape[0]), *[C.c_void_p(o.ptr) for o in out], None)
        if rc == _lib.ZK_ERR_LEN:
            raise ValueError((lib().zk_last_error() or b"").decode())
        check(rc)
        return tuple(o.to_numpy(np.uint64, (self.n_constraints, 4)) for o in out)

    def free(self):
        if self.handle.value:
            check(lib().zk_bn254_r1cs_free(self.handle))
            self.handle = C.c_uint64(0)


def setup(r1cs: R1CS, toxic, precompute_tables: bool = True):
    """groth16.Setup(r1cs) with the toxic waste (tau, alpha, beta, gamma, delta; (5, 4) Montgomery) as input -> (ProvingKey, vk dict)."""
    tx = np.ascontiguousarray(toxic, dtype=np.uint64).reshape(5, 4)
    h = C.c_uint64(0)
    vk1 = np.zeros((1 + r1cs.n_public, 8), np.uint64)
    vk2 = np.zeros((3, 16), np.uint64)
    check(lib().zk_bn254_groth16_setup(r1cs.handle, vp(tx), C.c_int(0 if precompute_tables else 1), C.byref(h), vp(vk1), vp(vk2)))
    log_n = max(r1cs.n_constraints - 1, 0).bit_length()
    pk = ProvingKey.from_handle(h.value, log_n, r1cs.n_wires, r1cs.n_public)
    return pk, dict(g1_alpha=vk1[0], g1_k=vk1[1:], g2_beta=vk2[0], g2_gamma=vk2[1], g2_delta=vk2[2])


def prove_r1cs(r1cs: R1CS, pk: ProvingKey, w, r, s) -> bytes:
    """groth16.Prove(r1cs, pk, witness): the full wire vector in (numpy, or a DeviceBuffer already in HBM), a / b / c on the device, 128 bytes out."""
    on_dev = isinstance(w, _lib.DeviceBuffer)
    if on_dev:
        wp, nw = C.c_void_p(w.ptr), r1cs.n_wires
    else:
        w = np.ascontiguousarray(w, dtype=np.uint64).reshape(-1, 4)
        wp, nw = vp(w), w.shape[0]
    r = np.ascontiguousarray(r, dtype=np.uint64).reshape(4)
    s = np.ascontiguousarray(s, dtype=np.uint64).reshape(4)
    proof = (C.c_uint8 * 128)()
    rc = lib().zk_bn254_groth16_prove_r1cs(r1cs.handle, pk.handle, wp, C.c_size_t(nw), vp(r), vp(s), C.c_int(int(on_dev)), proof)
    if rc == _lib.ZK_ERR_LEN:
        raise ValueError((lib().zk_last_error() or b"").decode())
    check(rc)
    return bytes(proof)
