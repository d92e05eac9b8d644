"""Host-side mirror of gnark-crypto v0.9.1's kzg package for the hot path (ecc/bn254/fr/kzg; pinned at /root/reference/gnark_backend_ffi/go.mod:5):
    kzg.NewSRS(size, alpha)         reached at /root/reference/gnark_backend_ffi/backend/common.go:137      -> new_srs(size, alpha)
    (*SRS).ReadFrom / WriteTo       what LoadSRS / SaveSRS move through srs.hex (backend/common.go:86-125) -> read_srs / SRS.write
    kzg.Commit(p, srs)              reached through plonk.Setup / plonk.Prove (backend/plonk/plonk.go:21,67) -> SRS.commit
The G1 side lives in HBM as a registered base array with its window tables; decoding a serialised SRS decompresses the points on the
device (one square root each) instead of on the host cores, and happens once instead of on every prove / verify call."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib, vp
from .bn254 import MultiExpConfig, ResidentBases


class SRS:
    def __init__(self, g1: ResidentBases, g2: np.ndarray, keep=None):
        self.g1, self.g2, self._keep = g1, g2, keep  # g2: (2, 16) uint64 = [G2, alpha * G2]

    @property
    def handle(self):
        return self.g1.handle

    def commit(self, poly, n: int | None = None) -> np.ndarray:
        """kzg.Commit: MultiExp(srs.G1[:len(p)], p); p = Montgomery coefficients (numpy (n, 4)) or a DeviceBuffer with n."""
        cfg = MultiExpConfig(scalars_mont=True)
        if isinstance(poly, _lib.DeviceBuffer):
            return self.g1.multi_exp_dev(poly, n, cfg)
        return self.g1.multi_exp(poly, cfg)

    def write(self, as_hex: bool = False) -> bytes:
        """(*SRS).WriteTo (as_hex: the text SaveSRS writes to srs.hex)"""
        nbytes = 132 + 32 * self.g1.n
        cap = 2 * nbytes if as_hex else nbytes
        buf = C.create_string_buffer(cap)
        n = C.c_size_t(0)
        check(lib().zk_bn254_kzg_srs_write(self.g1.handle, vp(np.ascontiguousarray(self.g2, dtype=np.uint64)), C.c_int(int(as_hex)), buf, C.c_size_t(cap), C.byref(n)))
        return buf.raw[:n.value]

    def free(self):
        self.g1.free()


def new_srs(size: int, alpha_mont, table_window_bits: int = 0) -> SRS:
    """kzg.NewSRS(size, alpha) on the device; alpha: Montgomery fr.Element (4 limbs)."""
    d = _lib.DeviceBuffer(max(size, 1) * 64)
    g2 = np.zeros((2, 16), np.uint64)
    check(lib().zk_bn254_kzg_new_srs_dev(C.c_void_p(d.ptr), C.c_size_t(size), vp(np.ascontiguousarray(alpha_mont, dtype=np.uint64)), vp(g2), None))
    rb = ResidentBases(d, n=size, table_window_bits=table_window_bits)
    d.free()
    return SRS(rb, g2)


def read_srs(data: bytes | str, is_hex: bool = False, table_window_bits: int = 0) -> SRS:
    """(*SRS).ReadFrom: bytes of WriteTo (or their hex text)."""
    raw = data.encode("ascii") if isinstance(data, str) else bytes(data)
    h, n = C.c_uint64(0), C.c_size_t(0)
    g2 = np.zeros((2, 16), np.uint64)
    rc = lib().zk_bn254_kzg_srs_read(C.c_char_p(raw), C.c_size_t(len(raw)), C.c_int(int(is_hex)), C.c_int(table_window_bits), C.byref(h), C.byref(n), vp(g2))
    if rc in (_lib.ZK_ERR_LEN, _lib.ZK_ERR_ARG):
        raise ValueError((lib().zk_last_error() or b"").decode())
    check(rc)
    rb = ResidentBases.__new__(ResidentBases)
    rb.is_g2, rb.handle, rb.n = False, h, int(n.value)
    return SRS(rb, g2)


def read_srs_g2(data: bytes | str, is_hex: bool = False) -> np.ndarray:
    """The two G2 points of an SRS image ([1]2, [alpha]2), decoded on the HOST (zk_bn254_kzg_srs_g2): all that plonk.Verify needs of the SRS; no device is touched."""
    raw = data.encode("ascii") if isinstance(data, str) else bytes(data)
    g2 = np.zeros((2, 16), np.uint64)
    rc = lib().zk_bn254_kzg_srs_g2(C.c_char_p(raw), C.c_size_t(len(raw)), C.c_int(int(is_hex)), vp(g2))
    if rc in (_lib.ZK_ERR_LEN, _lib.ZK_ERR_ARG):
        raise ValueError((lib().zk_last_error() or b"").decode())
    check(rc)
    return g2
