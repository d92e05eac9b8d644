"""Host-side mirror of the gnark-crypto v0.9.1 interface for the hot path (names and argument meaning follow upstream;
pinned at /root/reference/gnark_backend_ffi/go.mod:5 and reached through groth16.Prove main.go:131 / plonk.Prove
backend/plonk/plonk.go:67):

    ecc.MultiExpConfig{NbTasks, ScalarsMont}      -> MultiExpConfig
    (*G1Jac).MultiExp(points, scalars, config)    -> g1_multi_exp(points, scalars, config)   (affine result)
    (*G2Jac).MultiExp                             -> g2_multi_exp
    fft.NewDomain(m); (*Domain).FFT / FFTInverse  -> Domain(m).fft / .fft_inverse (in place, returns the array)
    fft.BitReverse                                -> bit_reverse
    fft.DIT / fft.DIF                             -> DIT / DIF

Containers are numpy uint64 arrays holding gnark's memory images: fr.Vector (n, 4); []G1Affine (n, 8);
[]G2Affine (n, 16) -- or `_lib.DeviceBuffer` / raw device pointers for data already resident in HBM.
Everything here dispatches to libzkmi.so; nothing is computed on the host."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import MsmCfg, check, lib, vp

DIT, DIF = 0, 1


@dataclass
class MultiExpConfig:
    nb_tasks: int = 0          # upstream NbTasks; > 1024 is an error like upstream, otherwise ignored on the GPU
    scalars_mont: bool = False  # upstream's zero value: scalars in regular form (gnark v0.8.0 calls FromMont() before MultiExp); True = Montgomery images
    window_bits: int = 0       # 0 = auto
    device_mask: int = 0       # several GPUs in one process (zk_init_devices): bit i = device entry i; 0 = the process default

    def _c(self) -> MsmCfg:
        return MsmCfg(self.nb_tasks, 1 if self.scalars_mont else 0, self.window_bits, self.device_mask)


def _as_u64(a, width) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.size % width:
        raise ValueError("array size %d is not a multiple of %d limbs" % (a.size, width))
    return a.reshape(-1, width)


def _multi_exp(fn, width, points, scalars, config):
    config = config or MultiExpConfig()
    points, scalars = _as_u64(points, width), _as_u64(scalars, 4)
    if points.shape[0] != scalars.shape[0]:
        # upstream: errors.New("len(points) != len(scalars)") -- raised by the library, not here, so that the C ABI is what is tested
        pass
    out = np.zeros(width, dtype=np.uint64)
    cfg = config._c()
    rc = fn(vp(points), C.c_size_t(points.shape[0]), vp(scalars), C.c_size_t(scalars.shape[0]), C.byref(cfg), vp(out))
    if rc in (_lib.ZK_ERR_LEN, _lib.ZK_ERR_NB_TASKS):
        raise ValueError((lib().zk_last_error() or b"").decode())
    check(rc)
    return out


def g1_multi_exp(points, scalars, config: MultiExpConfig | None = None) -> np.ndarray:
    """sum_i scalars[i] * points[i] on G1; returns the G1Affine image (8 limbs)."""
    return _multi_exp(lib().zk_bn254_g1_msm, 8, points, scalars, config)


def g2_multi_exp(points, scalars, config: MultiExpConfig | None = None) -> np.ndarray:
    """sum_i scalars[i] * points[i] on G2; returns the G2Affine image (16 limbs)."""
    return _multi_exp(lib().zk_bn254_g2_msm, 16, points, scalars, config)


def g1_multi_exp_dev(d_points: int, d_scalars: int, n: int, config: MultiExpConfig | None = None, stream: int = 0, partial=False):
    """Device-pointer variant (inputs resident in HBM).  partial=True returns the un-normalised XYZZ sum (16 limbs)."""
    cfg = (config or MultiExpConfig())._c()
    out = np.zeros(16 if partial else 8, dtype=np.uint64)
    fn = lib().zk_bn254_g1_msm_partial_dev if partial else lib().zk_bn254_g1_msm_dev
    check(fn(C.c_void_p(d_points), C.c_void_p(d_scalars), C.c_size_t(n), C.byref(cfg), vp(out), C.c_void_p(stream)))
    return out


def g2_multi_exp_dev(d_points: int, d_scalars: int, n: int, config: MultiExpConfig | None = None, stream: int = 0, partial=False):
    cfg = (config or MultiExpConfig())._c()
    out = np.zeros(32 if partial else 16, dtype=np.uint64)
    fn = lib().zk_bn254_g2_msm_partial_dev if partial else lib().zk_bn254_g2_msm_dev
    check(fn(C.c_void_p(d_points), C.c_void_p(d_scalars), C.c_size_t(n), C.byref(cfg), vp(out), C.c_void_p(stream)))
    return out


def g1_sum_partials(partials) -> np.ndarray:
    """Combine XYZZ partial sums (k, 16) from range-sharded MSMs into one affine point (host, O(k))."""
    p = _as_u64(partials, 16)
    out = np.zeros(8, dtype=np.uint64)
    check(lib().zk_bn254_g1_sum_xyzz(vp(p), C.c_size_t(p.shape[0]), vp(out)))
    return out


def g2_sum_partials(partials) -> np.ndarray:
    p = _as_u64(partials, 32)
    out = np.zeros(16, dtype=np.uint64)
    check(lib().zk_bn254_g2_sum_xyzz(vp(p), C.c_size_t(p.shape[0]), vp(out)))
    return out


class ResidentBases:
    """pk / SRS bases kept in HBM across calls (zk_bn254_bases_register)."""

    def __init__(self, points, is_g2: bool = False, n: int | None = None, table_window_bits: int = 0):
        """points: gnark memory images (numpy) -- or a DeviceBuffer / raw device pointer together with `n` (bases already in HBM).
        table_window_bits: 0 = the planner decides (window tables for >= 4096 bases), -1 = none, else the tables' window width."""
        self.is_g2, self.handle = is_g2, C.c_uint64(0)
        if isinstance(points, (_lib.DeviceBuffer, int)):
            if n is None:
                raise ValueError("n is required with device-resident points")
            self.n = n
            ptr = points.ptr if isinstance(points, _lib.DeviceBuffer) else int(points)
            check(lib().zk_bn254_bases_register_cfg(C.c_void_p(ptr), C.c_size_t(n), C.c_int(int(is_g2)), C.c_int(1), C.c_int(table_window_bits), C.byref(self.handle)))
            return
        points = _as_u64(points, 16 if is_g2 else 8)
        self.n = points.shape[0]
        check(lib().zk_bn254_bases_register_cfg(vp(points), C.c_size_t(self.n), C.c_int(int(is_g2)), C.c_int(0), C.c_int(table_window_bits), C.byref(self.handle)))

    def multi_exp_dev(self, d_scalars, n: int, config: MultiExpConfig | None = None, offset: int = 0) -> np.ndarray:
        """kzg.Commit of a polynomial that already lives in HBM (DeviceBuffer or raw device pointer)."""
        out = np.zeros(16 if self.is_g2 else 8, dtype=np.uint64)
        cfg = (config or MultiExpConfig())._c()
        ptr = d_scalars.ptr if isinstance(d_scalars, _lib.DeviceBuffer) else int(d_scalars)
        rc = lib().zk_bn254_msm_bases_dev(self.handle, C.c_size_t(offset), C.c_void_p(ptr), C.c_size_t(n), C.byref(cfg), vp(out))
        if rc in (_lib.ZK_ERR_LEN, _lib.ZK_ERR_NB_TASKS):
            raise ValueError((lib().zk_last_error() or b"").decode())
        check(rc)
        return out

    def multi_exp(self, scalars, config: MultiExpConfig | None = None, offset: int = 0) -> np.ndarray:
        scalars = _as_u64(scalars, 4)
        out = np.zeros(16 if self.is_g2 else 8, dtype=np.uint64)
        cfg = (config or MultiExpConfig())._c()
        rc = lib().zk_bn254_msm_bases(self.handle, C.c_size_t(offset), vp(scalars), C.c_size_t(scalars.shape[0]), C.byref(cfg), vp(out))
        if rc in (_lib.ZK_ERR_LEN, _lib.ZK_ERR_NB_TASKS):
            raise ValueError((lib().zk_last_error() or b"").decode())
        check(rc)
        return out

    def lagrange(self, log_n: int) -> "ResidentBases":
        """The Lagrange form of these G1 bases over the domain of 2^log_n points (zk_bn254_bases_lagrange): 2^log_n + 2 bases of their own."""
        rb = ResidentBases.__new__(ResidentBases)
        rb.is_g2, rb.handle, rb.n = False, C.c_uint64(0), (1 << log_n) + 2
        check(lib().zk_bn254_bases_lagrange(self.handle, C.c_uint32(log_n), C.byref(rb.handle)))
        return rb

    def build_table(self, table_window_bits: int = 0) -> None:
        """Window tables for bases registered without them (zk_bn254_bases_build_table); a no-op when they exist."""
        check(lib().zk_bn254_bases_build_table(self.handle, C.c_int(table_window_bits)))

    def multi_exp_batch(self, vectors, n: int | None = None, config: MultiExpConfig | None = None, offset: int = 0) -> np.ndarray:
        """Several scalar vectors of one length against these bases in one call (zk_bn254_msm_bases_batch[_dev]): numpy (n, 4) arrays, or DeviceBuffers / raw
        device pointers together with n.  Returns (len(vectors), 8 or 16) affine points -- the same as one multi_exp per vector."""
        cnt = len(vectors)
        out = np.zeros((cnt, 16 if self.is_g2 else 8), dtype=np.uint64)
        cfg = (config or MultiExpConfig())._c()
        on_dev = cnt > 0 and isinstance(vectors[0], (_lib.DeviceBuffer, int))
        if on_dev:
            if n is None:
                raise ValueError("n is required with device-resident scalars")
            ptrs = (C.c_void_p * cnt)(*[v.ptr if isinstance(v, _lib.DeviceBuffer) else int(v) for v in vectors])
            rc = lib().zk_bn254_msm_bases_batch_dev(self.handle, C.c_size_t(offset), ptrs, C.c_uint32(cnt), C.c_size_t(n), C.byref(cfg), vp(out))
        else:
            keep = [_as_u64(v, 4) for v in vectors]
            if any(k.shape[0] != keep[0].shape[0] for k in keep):
                raise ValueError("the scalar vectors of one batch have one length")
            ptrs = (C.c_void_p * cnt)(*[k.ctypes.data for k in keep])
            rc = lib().zk_bn254_msm_bases_batch(self.handle, C.c_size_t(offset), ptrs, C.c_uint32(cnt), C.c_size_t(keep[0].shape[0] if cnt else 0), C.byref(cfg), vp(out))
        if rc in (_lib.ZK_ERR_LEN, _lib.ZK_ERR_NB_TASKS):
            raise ValueError((lib().zk_last_error() or b"").decode())
        check(rc)
        return out

    def multi_exp_prepared(self, scalars: "PreparedScalars", skip: int = 0, offset: int = 0, config: MultiExpConfig | None = None) -> np.ndarray:
        """sum_{i >= skip} scalars[i] * bases[offset + i - skip] against scalars that were uploaded and are recoded once (zk_bn254_msm_bases_prepared)."""
        out = np.zeros(16 if self.is_g2 else 8, dtype=np.uint64)
        cfg = (config or MultiExpConfig())._c()
        rc = lib().zk_bn254_msm_bases_prepared(self.handle, C.c_size_t(offset), scalars.handle, C.c_size_t(skip), C.byref(cfg), vp(out))
        if rc in (_lib.ZK_ERR_LEN, _lib.ZK_ERR_NB_TASKS):
            raise ValueError((lib().zk_last_error() or b"").decode())
        check(rc)
        return out

    def free(self):
        if self.handle.value:
            lib().zk_bn254_bases_free(self.handle)
            self.handle = C.c_uint64(0)


class PreparedScalars:
    """A scalar vector uploaded once and recoded once per table geometry (zk_bn254_scalars_register): what groth16.Prove's A, B1, K and G2.B MultiExp calls share."""

    def __init__(self, scalars, config: MultiExpConfig | None = None):
        scalars = _as_u64(scalars, 4)
        self.n, self.handle = scalars.shape[0], C.c_uint64(0)
        cfg = (config or MultiExpConfig())._c()
        check(lib().zk_bn254_scalars_register(vp(scalars), C.c_size_t(self.n), C.byref(cfg), C.byref(self.handle)))

    def free(self):
        if self.handle.value:
            lib().zk_bn254_scalars_free(self.handle)
            self.handle = C.c_uint64(0)


class Domain:
    """fft.NewDomain(m): Cardinality = next power of two >= m.  The generator / coset / twiddle tables live on the device
    (built lazily per size by the library)."""

    def __init__(self, m: int):
        n = 1
        while n < m:
            n <<= 1
        self.cardinality = n
        self.log_n = n.bit_length() - 1
        if self.log_n > 28:
            raise ValueError("domain size 2^%d exceeds the Fr two-adicity 2^28" % self.log_n)

    def _run(self, a, inverse, decimation, coset):
        if decimation not in (DIT, DIF):
            raise ValueError("decimation must be DIT or DIF")
        if isinstance(a, (int, _lib.DeviceBuffer)):
            ptr = a if isinstance(a, int) else a.ptr
            check(lib().zk_bn254_ntt_dev(C.c_void_p(ptr), C.c_uint32(self.log_n), C.c_int(int(inverse)), C.c_int(decimation),
                                         C.c_int(int(bool(coset))), C.c_void_p(0)))
            return a
        if not (isinstance(a, np.ndarray) and a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]):
            raise TypeError("in-place transform needs a C-contiguous uint64 numpy array (or a device buffer)")
        if a.size != 4 * self.cardinality:
            raise ValueError("len(a) = %d != domain cardinality %d" % (a.size // 4, self.cardinality))
        check(lib().zk_bn254_ntt(vp(a), C.c_uint32(self.log_n), C.c_int(int(inverse)), C.c_int(decimation), C.c_int(int(bool(coset)))))
        return a

    def fft(self, a, decimation: int, coset: bool = False):
        """(*Domain).FFT(a, decimation, coset...) -- in place."""
        return self._run(a, False, decimation, coset)

    def fft_inverse(self, a, decimation: int, coset: bool = False):
        """(*Domain).FFTInverse(a, decimation, coset...) -- in place, includes the 1/N scaling."""
        return self._run(a, True, decimation, coset)


def bit_reverse(a):
    """fft.BitReverse(a) -- in place."""
    if not (isinstance(a, np.ndarray) and a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]):
        raise TypeError("needs a C-contiguous uint64 numpy array")
    n = a.size // 4
    log_n = n.bit_length() - 1
    if n == 0 or (1 << log_n) != n:
        raise ValueError("BitReverse needs a power-of-two length")
    check(lib().zk_bn254_bit_reverse(vp(a), C.c_uint32(log_n)))
    return a
