"""CPU suite: zk_bn254_kzg_srs_g2 -- the two G2 points of an SRS image decoded on the HOST (what the export shim's PlonkVerifyWithVK takes from srs.hex, so that
a process that only verifies never starts the HIP runtime; the reference re-reads the whole file for them: backend/plonk/plonk.go:34, backend/common.go:86-125).
Against the oracle's encoder (oracle/plonk_ref.kzg_srs_bytes) and its points; runs without a GPU and must not create a device entry."""
import ctypes as C

import pytest

from noir_backend_using_gnark_amd import _lib, kzg
from oracle import bn254_ref as ref
from oracle import plonk_ref as pl


def test_srs_g2_on_the_host_bytes_hex_and_errors():
    srs = pl.kzg_new_srs(12, 0xC0FFEE1234567, fast=True)
    wire = pl.kzg_srs_bytes(srs)
    want = [ref.g2_affine_mont_bytes(srs["g2"][k]) for k in range(2)]
    for data, is_hex in ((wire, False), (wire.hex(), True), (wire.hex().upper(), True)):
        g2 = kzg.read_srs_g2(data, is_hex=is_hex)
        assert g2[0].tobytes() == want[0] and g2[1].tobytes() == want[1]

    def bad(mutate, hexed=False):
        b = bytearray(wire)
        mutate(b)
        with pytest.raises(ValueError):
            kzg.read_srs_g2(bytes(b).hex() if hexed else bytes(b), is_hex=hexed)

    bad(lambda b: b.__setitem__(131, b[131] ^ 1))                    # the count does not match the length
    bad(lambda b: b.__setitem__(0, b[0] ^ 0x01), hexed=True)         # G2[0]: another x (no point / outside the r-torsion)
    bad(lambda b: b.__setitem__(64, b[64] & 0x3F))                   # G2[1]: flag 0b00, an uncompressed encoding in a compressed slot
    with pytest.raises(ValueError):
        kzg.read_srs_g2(wire[:-1])
    with pytest.raises(ValueError):
        kzg.read_srs_g2(wire.hex()[:9] + "g" + wire.hex()[10:], is_hex=True)  # not hex inside the header
    # the G1 points are not looked at (the reference ignores what ReadFrom makes of them when it only verifies): garbage there is not this function's business
    b = bytearray(wire)
    b[132:164] = b"\xff" * 32
    assert kzg.read_srs_g2(bytes(b))[1].tobytes() == want[1]
    assert _lib.lib().zk_device_entries(None, C.c_size_t(0)) == 0    # nothing above started the HIP runtime
