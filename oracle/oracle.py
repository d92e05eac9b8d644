"""ctypes binding of the C oracle (oracle/bn254_oracle.c) -- TEST INFRASTRUCTURE ONLY.

numpy conventions: Fr/Fp vectors are uint64 arrays of shape (n, 4) (gnark-crypto memory image: LE limbs,
Montgomery); G1 affine (n, 8); G2 affine (n, 16)."""
from __future__ import annotations
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "libbn254_oracle.so")
DIT, DIF = 0, 1


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("bn254_oracle.c", "bn254_oracle.h", "curve_tmpl.h", "plonk_oracle_impl.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "build/libbn254_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        try:
            _lib = C.CDLL(_SO)
        except OSError:
            build(force=True)
            _lib = C.CDLL(_SO)
        _u64p = C.POINTER(C.c_uint64)
        _lib.orc_max_threads.restype = C.c_int
        for name in ("orc_g1_msm", "orc_g2_msm", "orc_g1_msm_naive", "orc_g2_msm_naive", "orc_groth16_prove",
                     "orc_g1_on_curve", "orc_g2_on_curve"):
            getattr(_lib, name).restype = C.c_int
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _u64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


def max_threads() -> int:
    """Threads the oracle starts: min(OpenMP's default, the cgroup CPU quota) -- the CPUs this process can actually use."""
    return int(lib().orc_max_threads())


def host_cpus() -> int:
    """Logical CPUs the box shows (the GPU boxes: 256 under a quota of 16)."""
    return int(lib().orc_host_cpus())


def int_to_limbs(x: int) -> np.ndarray:
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def limbs_to_int(a) -> int:
    a = np.asarray(a, dtype=np.uint64).reshape(-1)
    return sum(int(a[i]) << (64 * i) for i in range(a.size))


def ints_to_limbs(xs) -> np.ndarray:
    out = np.zeros((len(xs), 4), dtype=np.uint64)
    for i, x in enumerate(xs):
        out[i] = int_to_limbs(x)
    return out


def fe_op(op: str, which: int, a, b=None) -> np.ndarray:
    out = np.zeros(4, dtype=np.uint64)
    a = _u64(a)
    fn = getattr(lib(), "orc_fe_" + op)
    if b is None:
        fn(C.c_int(which), _p(a), _p(out))
    else:
        fn(C.c_int(which), _p(a), _p(_u64(b)), _p(out))
    return out


def to_mont_vec(xs, which: int = 0) -> np.ndarray:
    """list of python ints (canonical) -> (n,4) Montgomery limbs"""
    can = ints_to_limbs(xs)
    out = np.zeros_like(can)
    for i in range(len(xs)):
        lib().orc_fe_to_mont(C.c_int(which), _p(can[i]), _p(out[i]))
    return out


def from_mont_vec(a, which: int = 0):
    a = _u64(a).reshape(-1, 4)
    out = []
    t = np.zeros(4, dtype=np.uint64)
    for i in range(a.shape[0]):
        lib().orc_fe_from_mont(C.c_int(which), _p(np.ascontiguousarray(a[i])), _p(t))
        out.append(limbs_to_int(t))
    return out


def rand_fr(seed: int, n: int, mont: bool = True, witness_like: bool = False) -> np.ndarray:
    out = np.zeros((n, 4), dtype=np.uint64)
    fn = lib().orc_rand_fr_witness if witness_like else lib().orc_rand_fr
    fn(C.c_uint64(seed), C.c_size_t(n), _p(out), C.c_int(1 if mont else 0))
    return out


def g1_gen_points(seed: int, n: int, nthreads: int = 0) -> np.ndarray:
    out = np.zeros((n, 8), dtype=np.uint64)
    lib().orc_g1_gen_points(C.c_uint64(seed), C.c_size_t(n), _p(out), C.c_int(nthreads or max_threads()))
    return out


def g2_gen_points(seed: int, n: int, nthreads: int = 0) -> np.ndarray:
    out = np.zeros((n, 16), dtype=np.uint64)
    lib().orc_g2_gen_points(C.c_uint64(seed), C.c_size_t(n), _p(out), C.c_int(nthreads or max_threads()))
    return out


def g1_on_curve(p) -> bool:
    return bool(lib().orc_g1_on_curve(_p(_u64(p).reshape(8))))


def g2_on_curve(p) -> bool:
    return bool(lib().orc_g2_on_curve(_p(_u64(p).reshape(16))))


def _msm(fn, width, points, scalars, scalars_mont, c, nthreads):
    points, scalars = _u64(points), _u64(scalars)
    n = scalars.size // 4
    assert points.size == n * width
    out = np.zeros(width, dtype=np.uint64)
    rc = fn(_p(points), _p(scalars), C.c_size_t(n), C.c_int(1 if scalars_mont else 0), C.c_int(c),
            C.c_int(nthreads or max_threads()), _p(out))
    assert rc == 0
    return out


def g1_msm(points, scalars, scalars_mont=True, c=0, nthreads=0):
    return _msm(lib().orc_g1_msm, 8, points, scalars, scalars_mont, c, nthreads)


def g2_msm(points, scalars, scalars_mont=True, c=0, nthreads=0):
    return _msm(lib().orc_g2_msm, 16, points, scalars, scalars_mont, c, nthreads)


def _msm_naive(fn, width, points, scalars, scalars_mont):
    points, scalars = _u64(points), _u64(scalars)
    n = scalars.size // 4
    out = np.zeros(width, dtype=np.uint64)
    fn(_p(points), _p(scalars), C.c_size_t(n), C.c_int(1 if scalars_mont else 0), _p(out))
    return out


def g1_msm_naive(points, scalars, scalars_mont=True):
    return _msm_naive(lib().orc_g1_msm_naive, 8, points, scalars, scalars_mont)


def g2_msm_naive(points, scalars, scalars_mont=True):
    return _msm_naive(lib().orc_g2_msm_naive, 16, points, scalars, scalars_mont)


def g1_add(a, b):
    out = np.zeros(8, dtype=np.uint64); lib().orc_g1_add(_p(_u64(a)), _p(_u64(b)), _p(out)); return out


def g2_add(a, b):
    out = np.zeros(16, dtype=np.uint64); lib().orc_g2_add(_p(_u64(a)), _p(_u64(b)), _p(out)); return out


def g1_mul(a, k, k_mont=True):
    out = np.zeros(8, dtype=np.uint64); lib().orc_g1_mul(_p(_u64(a)), _p(_u64(k)), C.c_int(int(k_mont)), _p(out)); return out


def g2_mul(a, k, k_mont=True):
    out = np.zeros(16, dtype=np.uint64); lib().orc_g2_mul(_p(_u64(a)), _p(_u64(k)), C.c_int(int(k_mont)), _p(out)); return out


def g1_compress(a) -> bytes:
    out = (C.c_uint8 * 32)(); lib().orc_g1_compress(_p(_u64(a)), out); return bytes(out)


def g2_compress(a) -> bytes:
    out = (C.c_uint8 * 64)(); lib().orc_g2_compress(_p(_u64(a)), out); return bytes(out)


def fr_ntt(a, inverse: bool, decimation: int, coset: bool = False, nthreads: int = 0, inplace: bool = False) -> np.ndarray:
    a = _u64(a) if inplace else _u64(a).copy()
    n = a.size // 4
    logn = n.bit_length() - 1
    assert 1 << logn == n
    lib().orc_fr_ntt(_p(a), C.c_uint(logn), C.c_int(int(inverse)), C.c_int(decimation), C.c_int(int(coset)),
                     C.c_int(nthreads or max_threads()))
    return a


def fr_bit_reverse(a) -> np.ndarray:
    a = _u64(a).copy()
    n = a.size // 4
    lib().orc_fr_bit_reverse(_p(a), C.c_uint(n.bit_length() - 1))
    return a


def groth16_compute_h(a, b, c, log_n: int, nthreads: int = 0) -> np.ndarray:
    a, b, c = _u64(a), _u64(b), _u64(c)
    n = a.size // 4
    h = np.zeros((1 << log_n, 4), dtype=np.uint64)
    lib().orc_groth16_compute_h(_p(a), _p(b), _p(c), C.c_size_t(n), C.c_uint(log_n), _p(h), C.c_int(nthreads or max_threads()))
    return h


class _PK(C.Structure):
    _fields_ = [("log_domain", C.c_uint), ("n_wires", C.c_size_t), ("n_public", C.c_size_t)] + \
               [(k, C.POINTER(C.c_uint64)) for k in ("g1_alpha", "g1_beta", "g1_delta", "g1_a", "g1_b", "g1_k", "g1_z",
                                                      "g2_beta", "g2_delta", "g2_b")]


def groth16_prove(pk: dict, a, b, c, w, r, s, nthreads: int = 0):
    """pk: dict of numpy arrays (keys as in orc_groth16_pk) + log_domain, n_wires, n_public.
    Returns (proof_bytes[128], points[32 limbs])."""
    keep = {k: _u64(pk[k]) for k in ("g1_alpha", "g1_beta", "g1_delta", "g1_a", "g1_b", "g1_k", "g1_z", "g2_beta", "g2_delta", "g2_b")}
    s_pk = _PK(pk["log_domain"], pk["n_wires"], pk["n_public"], *[_p(keep[k]) if keep[k].size else None for k in
               ("g1_alpha", "g1_beta", "g1_delta", "g1_a", "g1_b", "g1_k", "g1_z", "g2_beta", "g2_delta", "g2_b")])
    a, b, c, w = _u64(a), _u64(b), _u64(c), _u64(w)
    proof = (C.c_uint8 * 128)()
    pts = np.zeros(32, dtype=np.uint64)
    rc = lib().orc_groth16_prove(C.byref(s_pk), _p(a), _p(b), _p(c), C.c_size_t(a.size // 4), _p(w), _p(_u64(r)), _p(_u64(s)),
                                 C.c_int(nthreads or max_threads()), proof, _p(pts))
    assert rc == 0
    return bytes(proof), pts


# ------------------------------------------------------------------------------------------------ PLONK (oracle/plonk_oracle_impl.h)
class _PlonkCircuit(C.Structure):
    _fields_ = [("n_public", C.c_size_t), ("n_constraints", C.c_size_t), ("n_vars", C.c_size_t)] + \
               [(k, C.POINTER(C.c_uint64)) for k in ("ql", "qr", "qm", "qo", "qk")] + \
               [(k, C.POINTER(C.c_uint32)) for k in ("xa", "xb", "xc")] + \
               [("srs_g1", C.POINTER(C.c_uint64)), ("srs_len", C.c_size_t)]


class PlonkKeyC:
    """plonk.Setup's key held by the C oracle (orc_plonk_setup): canonical selector / permutation polynomials + the verifying key's digests."""
    NAMES = ("ql", "qr", "qm", "qo", "cqk", "lqk", "s1", "s2", "s3")

    def __init__(self, n_public, n_vars, ql, qr, qm, qo, qk, xa, xb, xc, srs_g1, nthreads: int = 0):
        """ql .. qk: (n_constraints, 4) Montgomery Fr; xa, xb, xc: uint32 variable indices; srs_g1: (>= n + 3, 8) affine points."""
        L = lib()
        L.orc_plonk_setup.restype = C.c_void_p
        L.orc_plonk_prove.restype = C.c_int
        co = [_u64(v).reshape(-1, 4) for v in (ql, qr, qm, qo, qk)]
        xs = [np.ascontiguousarray(v, dtype=np.uint32) for v in (xa, xb, xc)]
        self._srs = _u64(srs_g1).reshape(-1, 8)  # borrowed by the C side: kept alive here
        nc = xs[0].size
        assert all(c.shape[0] == nc for c in co) and all(x.size == nc for x in xs)
        u32p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
        circ = _PlonkCircuit(n_public, nc, n_vars, *[_p(c) for c in co], *[u32p(x) for x in xs], _p(self._srs), self._srs.shape[0])
        self.nthreads = nthreads or max_threads()
        self.h = L.orc_plonk_setup(C.byref(circ), C.c_int(self.nthreads))
        if not self.h:
            raise ValueError("orc_plonk_setup refused the circuit (empty system, a variable index out of range, or an SRS shorter than n + 3)")
        n, n4 = C.c_size_t(0), C.c_size_t(0)
        L.orc_plonk_pk_sizes(C.c_void_p(self.h), C.byref(n), C.byref(n4))
        self.n, self.n4, self.n_public = n.value, n4.value, n_public

    def poly(self, name: str) -> np.ndarray:
        out = np.zeros((self.n, 4), dtype=np.uint64)
        lib().orc_plonk_pk_get(C.c_void_p(self.h), C.c_int(self.NAMES.index(name)), out.ctypes.data_as(C.c_void_p))
        return out

    def vk_digests(self) -> np.ndarray:
        """[S1] [S2] [S3] [Ql] [Qr] [Qm] [Qo] [Qk] as (8, 8) affine Montgomery limbs."""
        out = np.zeros((8, 8), dtype=np.uint64)
        lib().orc_plonk_pk_get(C.c_void_p(self.h), C.c_int(9), out.ctypes.data_as(C.c_void_p))
        return out

    def perm(self) -> np.ndarray:
        out = np.zeros(3 * self.n, dtype=np.uint32)
        lib().orc_plonk_pk_get(C.c_void_p(self.h), C.c_int(10), out.ctypes.data_as(C.c_void_p))
        return out

    def prove(self, solution, blinders, want_challenges: bool = False):
        """solution: (n_vars, 4) Montgomery; blinders: (9, 4) Montgomery.  -> the 548 bytes of Proof.WriteTo (and the five challenges as ints)."""
        sol, bl = _u64(solution).reshape(-1, 4), _u64(blinders).reshape(9, 4)
        proof = (C.c_uint8 * 548)()
        ch = np.zeros((5, 4), dtype=np.uint64)
        rc = lib().orc_plonk_prove(C.c_void_p(self.h), _p(sol), _p(bl), C.c_int(self.nthreads), proof, _p(ch))
        if rc == -2:
            raise AssertionError("the constraint system is not satisfied (the quotient is not a polynomial)")
        assert rc == 0, rc
        if want_challenges:
            return bytes(proof), dict(zip(("gamma", "beta", "alpha", "zeta", "kzg_gamma"), from_mont_vec(ch)))
        return bytes(proof)

    def free(self):
        if self.h:
            lib().orc_plonk_pk_free(C.c_void_p(self.h))
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sha256(data: bytes) -> bytes:
    out = (C.c_uint8 * 32)()
    lib().orc_sha256((C.c_uint8 * len(data)).from_buffer_copy(data) if data else None, C.c_size_t(len(data)), out)
    return bytes(out)
