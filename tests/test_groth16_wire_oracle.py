"""CPU suite: the oracle's restatement of gnark's Groth16 key wire formats (ProvingKey.WriteTo / ReadFrom, VerifyingKey.WriteTo -- the hex payloads
of the reference's intended Groth16 FFI, backend/groth16/r1cs.go:107-143, 214-266) against the committed fixtures, which are the keys behind the
pairing-verified golden proofs."""
import hashlib
import json
import os

import pytest

from oracle import bn254_ref as ref
from oracle import plonk_ref as pl
from tests.helpers import golden_pk, h2i

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def wire():
    with open(os.path.join(HERE, "golden", "groth16_wire_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "bn254_golden.json")) as f:
        return json.load(f)


def test_pk_bytes_layout_and_round_trip(wire):
    for e in wire:
        b = bytes.fromhex(e["pk_hex"])
        assert hashlib.sha256(b).hexdigest() == e["pk_sha256"]
        pk = pl.groth16_pk_from_bytes(b)
        nw = e["n_wires"]
        assert len(pk["infinity_a"]) == nw and sum(pk["infinity_a"]) == e["nb_infinity_a"] and sum(pk["infinity_b"]) == e["nb_infinity_b"]
        assert len(pk["g1_a"]) == nw - e["nb_infinity_a"] and len(pk["g1_b"]) == len(pk["g2_b"]) == nw - e["nb_infinity_b"]
        assert len(pk["g1_k"]) == nw - e["n_public"] and len(pk["g1_z"]) == pk["domain"].n
        assert pl.groth16_pk_bytes(pk) == b
        # the layout, field by field: 168 (domain) + 96 + four G1 slices + 128 + the G2 slice + 24 + 2 * nbWires
        na, nb = len(pk["g1_a"]), len(pk["g1_b"])
        assert len(b) == 168 + 96 + 4 * 4 + 32 * (na + nb + pk["domain"].n + len(pk["g1_k"])) + 128 + 4 + 64 * nb + 24 + 2 * nw
        assert int.from_bytes(b[0:8], "big") == pk["domain"].n and int.from_bytes(b[264:268], "big") == na
        assert b[-2 * nw:-nw] == bytes(pk["infinity_a"]) and b[-nw:] == bytes(pk["infinity_b"])


def test_decoded_key_reproduces_the_golden_proofs(wire, golden):
    """ReadFrom -> expand -> prove gives the committed (pairing-verified) proof bytes: the wire image holds exactly the key those proofs were made with."""
    by_name = {e["name"]: e for e in golden["groth16"]}
    for e in wire:
        g = by_name[e["name"]]
        pk = pl.groth16_pk_expand(pl.groth16_pk_from_bytes(bytes.fromhex(e["pk_hex"])))
        a, b, c, w = ([h2i(v) for v in g[k]] for k in ("a", "b", "c", "w"))
        proof = ref.groth16_prove(pk, g["n_public"], a, b, c, w, h2i(g["r"]), h2i(g["s"]))
        assert ref.groth16_proof_bytes(*proof).hex() == g["proof"]
        # and it is the key of the fixture point for point (Montgomery images there)
        gp = golden_pk(g)
        assert b"".join(ref.g1_affine_mont_bytes(P) for P in pk["g1_a"]) == gp["g1_a"].tobytes()
        assert b"".join(ref.g2_affine_mont_bytes(P) for P in pk["g2_b"]) == gp["g2_b"].tobytes()
        assert b"".join(ref.g1_affine_mont_bytes(P) for P in pk["g1_z"]) == gp["g1_z"].tobytes()


def test_vk_bytes(wire):
    for e in wire:
        b = bytes.fromhex(e["vk_hex"])
        assert len(b) == 292 + 32 * e["n_public"] and int.from_bytes(b[288:292], "big") == e["n_public"]
        pkb = bytes.fromhex(e["pk_hex"])
        assert b[0:32] == pkb[168:200] and b[32:64] == pkb[200:232] and b[192:224] == pkb[232:264]  # [alpha]1, [beta]1, [delta]1 are the proving key's
        for o in (64, 128, 224):
            assert pl.g2_decompress(b[o:o + 64]) is not None


def test_pk_decoder_rejects_malformed_keys(wire):
    good = bytes.fromhex(wire[1]["pk_hex"])
    nw = wire[1]["n_wires"]

    def bad(mut):
        b = bytearray(good)
        mut(b)
        with pytest.raises(ValueError):
            pl.groth16_pk_from_bytes(bytes(b))

    with pytest.raises(ValueError):
        pl.groth16_pk_from_bytes(good[:-1])
    with pytest.raises(ValueError):
        pl.groth16_pk_from_bytes(good + b"\0")
    bad(lambda b: b.__setitem__(7, 17))                       # cardinality not a power of two
    bad(lambda b: b.__setitem__(40, b[40] ^ 1))               # another generator
    bad(lambda b: b.__setitem__(168, b[168] & 0x3F))          # uncompressed flag inside a compressed stream
    bad(lambda b: b.__setitem__(slice(168, 200), b"\xbf" + b"\xff" * 31))  # x >= q
    bad(lambda b: b.__setitem__(len(b) - 1, 2))               # a bool that is neither 0 nor 1
    bad(lambda b: b.__setitem__(len(b) - 2 * nw - 1, b[len(b) - 2 * nw - 1] ^ 1))  # NbInfinityB does not match the bitmap
    bad(lambda b: b.__setitem__(267, b[267] + 1))             # len(A) does not match


def test_g2_decompress_rejects_non_residue_and_accepts_both_signs():
    P = ref.g2_mul(ref.G2_GEN, 12345)
    enc = ref.g2_compress(P)
    assert pl.g2_decompress(enc) == P
    neg = (P[0], ref.f2_neg(P[1]))
    assert pl.g2_decompress(ref.g2_compress(neg)) == neg and ref.g2_compress(neg)[0] >> 6 != enc[0] >> 6
    # an x with no point on the twist
    x = 5
    while True:
        rhs = ref.f2_add(ref.f2_mul(ref.f2_sqr((x, 1)), (x, 1)), ref.B_G2)
        if pl.f2_sqrt(rhs) is None:
            break
        x += 1
    with pytest.raises(ValueError):
        pl.g2_decompress(bytes([0x80]) + (1).to_bytes(31, "big") + x.to_bytes(32, "big"))


def test_g2_decompress_applies_the_subgroup_check():
    """x = 2 + u is on the twist but outside the r-torsion: the Decoder's default subgroup check rejects it."""
    X = (2, 1)
    y = pl.f2_sqrt(ref.f2_add(ref.f2_mul(ref.f2_sqr(X), X), ref.B_G2))
    assert y is not None and ref.g2_on_curve((X, y))
    enc = ref.g2_compress((X, y))
    with pytest.raises(ValueError):
        pl.g2_decompress(enc)
    assert pl.g2_decompress(enc, subgroup_check=False) == (X, y)
