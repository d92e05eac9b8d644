"""GPU parity suite: Groth16 key wire formats on the device and the Groth16 FFI entry points that move them (row f2) --
zk_bn254_groth16_pk_read / _pk_write / _vk_write, zk_groth16_preprocess / _prove_with_pk / _prove_with_meta -- against the oracle's restatement
of gnark v0.8.0's ProvingKey.WriteTo / ReadFrom (oracle/plonk_ref.py) and the committed fixtures."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from noir_backend_using_gnark_amd import _lib
from noir_backend_using_gnark_amd import groth16 as zk
from noir_backend_using_gnark_amd._lib import vp
from oracle import bn254_ref as ref
from oracle import oracle as orc
from oracle import plonk_ref as pl
from tests.helpers import golden_pk, h2i, mont_limbs

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def wire():
    with open(os.path.join(HERE, "golden", "groth16_wire_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "bn254_golden.json")) as f:
        return json.load(f)


def _dense_key(g):
    pkd = golden_pk(g)
    return zk.ProvingKey(pkd["log_domain"], pkd["n_wires"], pkd["n_public"], pkd["g1_alpha"], pkd["g1_beta"], pkd["g1_delta"], pkd["g1_a"], pkd["g1_b"],
                         pkd["g1_k"], pkd["g1_z"], pkd["g2_beta"], pkd["g2_delta"], pkd["g2_b"])


def test_pk_write_and_read_on_the_golden_keys(wire, golden):
    """WriteTo of the resident golden keys == the committed wire images (bytes and hex); ReadFrom of those images proves the committed, pairing-verified
    proof bytes -- with and without window tables."""
    by_name = {e["name"]: e for e in golden["groth16"]}
    for e in wire:
        g = by_name[e["name"]]
        pk = _dense_key(g)
        assert pk.write_to().hex() == e["pk_hex"]
        assert pk.write_to(as_hex=True) == e["pk_hex"]
        pk.free()
        a, b, c, w = (mont_limbs([h2i(v) for v in g[k]]) for k in ("a", "b", "c", "w"))
        r, s = mont_limbs([h2i(g["r"])])[0], mont_limbs([h2i(g["s"])])[0]
        for src, tables in ((e["pk_hex"], True), (bytes.fromhex(e["pk_hex"]), False), (e["pk_hex"].upper(), True)):
            rk = zk.ProvingKey.read_from(src, is_hex=isinstance(src, str), precompute_tables=tables)
            assert (rk.n_wires, rk.n_public, rk.log_domain) == (g["n_wires"], g["n_public"], golden_pk(g)["log_domain"])
            assert zk.prove(rk, a, b, c, w, r, s).hex() == g["proof"]
            assert rk.write_to().hex() == e["pk_hex"]
            rk.free()


def _random_key(log_n, seed=0):
    N = 1 << log_n
    nw, npub = N - 3, 5
    pkd = dict(log_domain=log_n, n_wires=nw, n_public=npub,
               g1_alpha=orc.g1_gen_points(seed + 1, 1)[0], g1_beta=orc.g1_gen_points(seed + 2, 1)[0], g1_delta=orc.g1_gen_points(seed + 3, 1)[0],
               g1_a=orc.g1_gen_points(seed + 4, nw), g1_b=orc.g1_gen_points(seed + 5, nw), g1_k=orc.g1_gen_points(seed + 6, nw - npub),
               g1_z=orc.g1_gen_points(seed + 7, N), g2_beta=orc.g2_gen_points(seed + 8, 1)[0], g2_delta=orc.g2_gen_points(seed + 9, 1)[0],
               g2_b=orc.g2_gen_points(seed + 10, nw))
    for i in (0, 7, nw - 1):
        pkd["g1_a"][i] = 0
    for i in (3, 11, 12, nw - 2):
        pkd["g1_b"][i] = 0
        pkd["g2_b"][i] = 0
    return pkd


def _oracle_key_bytes(pkd):
    """the oracle's WriteTo of a key given as Montgomery images"""
    from tests.helpers import from_mont_limbs

    def g1(a):
        v = from_mont_limbs(np.asarray(a, np.uint64).reshape(-1, 4), ref.Q)
        return [None if (v[2 * i] | v[2 * i + 1]) == 0 else (v[2 * i], v[2 * i + 1]) for i in range(len(v) // 2)]

    def g2(a):
        v = from_mont_limbs(np.asarray(a, np.uint64).reshape(-1, 4), ref.Q)
        return [None if not any(v[4 * i:4 * i + 4]) else ((v[4 * i], v[4 * i + 1]), (v[4 * i + 2], v[4 * i + 3])) for i in range(len(v) // 4)]

    opk = dict(domain=ref.Domain(1 << pkd["log_domain"]), g1_alpha=g1(pkd["g1_alpha"])[0], g1_beta=g1(pkd["g1_beta"])[0], g1_delta=g1(pkd["g1_delta"])[0],
               g1_a=g1(pkd["g1_a"]), g1_b=g1(pkd["g1_b"]), g1_k=g1(pkd["g1_k"]), g1_z=g1(pkd["g1_z"]), g2_beta=g2(pkd["g2_beta"])[0],
               g2_delta=g2(pkd["g2_delta"])[0], g2_b=g2(pkd["g2_b"]))
    return pl.groth16_pk_bytes(opk)


def test_pk_wire_format_random_key_2p10_vs_oracle():
    """A 2^10-constraint key with points at infinity in A and B: the device's WriteTo equals the oracle's byte for byte; the device's ReadFrom of the
    oracle's bytes proves the oracle's proof bytes (every decompressed point -- 3 x 1021 G1 square roots, 1017 Fp2 square roots with their subgroup
    checks -- is the original one, or the proof would differ)."""
    log_n = 10
    pkd = _random_key(log_n)
    want = _oracle_key_bytes(pkd)
    pk = zk.ProvingKey(**pkd)
    assert pk.write_to() == want
    pk.free()
    N, nw = 1 << log_n, pkd["n_wires"]
    a, b = orc.rand_fr(20, N - 10), orc.rand_fr(21, N - 10)
    c = np.stack([orc.fe_op("mul", 0, a[i], b[i]) for i in range(N - 10)])
    w = orc.rand_fr(22, nw)
    r, s = orc.rand_fr(23, 1)[0], orc.rand_fr(24, 1)[0]
    exp, _ = orc.groth16_prove(pkd, a, b, c, w, r, s)
    rk = zk.ProvingKey.read_from(want)
    assert rk.info() == dict(n_wires=nw, n_public=5, log_domain=log_n, tables=True)
    assert zk.prove(rk, a, b, c, w, r, s) == exp
    assert rk.write_to(as_hex=True) == want.hex()
    rk.free()


def test_pk_read_rejects_malformed_keys(wire):
    """Every failure of gnark's ReadFrom / gnark-crypto's Decoder has its error here: truncation, a non-hex character, flag bits, coordinates >= q,
    x without a point, a G2 point outside the r-torsion, inconsistent counts, a foreign domain."""
    good = bytes.fromhex(wire[1]["pk_hex"])
    nw = wire[1]["n_wires"]
    opk = pl.groth16_pk_from_bytes(good)
    na, nb = len(opk["g1_a"]), len(opk["g1_b"])
    at_a = 268
    at_g2b = at_a + 32 * na + 4 + 32 * nb + 4 + 32 * opk["domain"].n + 4 + 32 * len(opk["g1_k"]) + 128 + 4

    def bad(mut, hexed=False):
        b = bytearray(good)
        mut(b)
        with pytest.raises(ValueError):
            zk.ProvingKey.read_from(bytes(b).hex() if hexed else bytes(b), is_hex=hexed)

    for cut in (1, 2 * nw, 2 * nw + 24, 500, len(good) - 100):
        with pytest.raises(ValueError):
            zk.ProvingKey.read_from(good[:-cut], is_hex=False)
    with pytest.raises(ValueError):
        zk.ProvingKey.read_from(good + b"\0", is_hex=False)
    hx = good.hex()
    for pos in (3, 700, len(hx) - 5):  # header field, a point, a bitmap byte
        with pytest.raises(ValueError):
            zk.ProvingKey.read_from(hx[:pos] + "g" + hx[pos + 1:], is_hex=True)
    with pytest.raises(ValueError):
        zk.ProvingKey.read_from(hx[:-1], is_hex=True)
    bad(lambda b: b.__setitem__(7, 17))
    bad(lambda b: b.__setitem__(40, b[40] ^ 1))
    bad(lambda b: b.__setitem__(168, b[168] & 0x3F))
    bad(lambda b: b.__setitem__(slice(168, 200), b"\xbf" + b"\xff" * 31), hexed=True)
    bad(lambda b: b.__setitem__(len(b) - 1, 2))
    bad(lambda b: b.__setitem__(len(b) - 2 * nw - 1, b[len(b) - 2 * nw - 1] ^ 1))
    bad(lambda b: b.__setitem__(267, b[267] + 1))
    # a G1 x with no point on the curve (x = 4: 67 is a non-residue? search) in A
    x = 1
    while pow((x ** 3 + 3) % ref.Q, (ref.Q - 1) // 2, ref.Q) == 1:
        x += 1
    bad(lambda b: b.__setitem__(slice(at_a, at_a + 32), bytes([0x80]) + x.to_bytes(31, "big")))
    # G2.B[0]: an x with no point on the twist; then a twist point outside the r-torsion (x = 2 + u)
    xx = 5
    while pl.f2_sqrt(ref.f2_add(ref.f2_mul(ref.f2_sqr((xx, 1)), (xx, 1)), ref.B_G2)) is not None:
        xx += 1
    bad(lambda b: b.__setitem__(slice(at_g2b, at_g2b + 64), bytes([0x80]) + (1).to_bytes(31, "big") + xx.to_bytes(32, "big")))
    X = (2, 1)
    y = pl.f2_sqrt(ref.f2_add(ref.f2_mul(ref.f2_sqr(X), X), ref.B_G2))
    bad(lambda b: b.__setitem__(slice(at_g2b, at_g2b + 64), ref.g2_compress((X, y))))
    bad(lambda b: b.__setitem__(slice(at_g2b, at_g2b + 64), bytes([0x40, 1]) + bytes(62)))  # infinity flag with a non-zero payload
    # the untouched key still loads after all that
    zk.ProvingKey.read_from(good, is_hex=False).free()


def test_pk_round_trip_2p16_and_proof():
    """Size: a 2^16-constraint key (65533 wires: ~196 k G1 + 65 k G2 decompressions) written and read back on the device proves the same bytes as the
    original resident key, and the second write equals the first."""
    log_n = 16
    pkd = _random_key(log_n, seed=40)
    N, nw = 1 << log_n, pkd["n_wires"]
    a, b = orc.rand_fr(20, N), orc.rand_fr(21, N)
    c = np.stack([orc.fe_op("mul", 0, a[i], b[i]) for i in range(N)])
    w = orc.rand_fr(22, nw, witness_like=True)
    r, s = orc.rand_fr(23, 1)[0], orc.rand_fr(24, 1)[0]
    pk = zk.ProvingKey(**pkd)
    want = zk.prove(pk, a, b, c, w, r, s)
    img = pk.write_to(as_hex=True)
    pk.free()
    assert len(img) == 2 * (168 + 96 + 16 + 32 * ((nw - 3) + (nw - 4) + N + (nw - 5)) + 128 + 4 + 64 * (nw - 4) + 24 + 2 * nw)
    rk = zk.ProvingKey.read_from(img)
    assert zk.prove(rk, a, b, c, w, r, s) == want
    assert rk.write_to(as_hex=True) == img
    rk.free()


def test_groth16_ffi_preprocess_prove_with_pk_and_with_meta():
    """The reference's intended Groth16 FFI on its RawR1CS payload: Preprocess -> hex keys equal to the oracle's WriteTo of the oracle's Setup (same
    toxic waste); ProveWithPK on that hex key and ProveWithMeta -> the oracle's proof bytes, accepted by the oracle's pairing verifier; with drawn
    randomness the proof still verifies."""
    from noir_backend_using_gnark_amd import frontend as fe
    hx = lambda v: "%064x" % (v % ref.R)
    w1, w2 = 7, 11
    w3 = w1 * w2 % ref.R
    w4 = (2 * w3 * w1 + 3 * w2 + 5) % ref.R
    values = [w1, w2, w3, w4, 123456789]
    raw = {"gates": [{"mul_terms": [{"coefficient": hx(1), "multiplicand": 1, "multiplier": 2}], "add_terms": [{"coefficient": hx(-1), "sum": 3}], "constant_term": hx(0)},
                     {"mul_terms": [{"coefficient": hx(2), "multiplicand": 3, "multiplier": 1}],
                      "add_terms": [{"coefficient": hx(3), "sum": 2}, {"coefficient": hx(-1), "sum": 4}], "constant_term": hx(5)}],
           "public_inputs": [4, 2], "values": ref.felts_wire(values).hex(), "num_variables": 6, "num_constraints": 2}
    rj = json.dumps(raw)
    r1, wv = pl.r1cs_from_raw(raw)
    a, b, c = r1.eval_abc(wv)
    tox, rs = tuple(ref.rand_felts(0xB0, 5)), tuple(ref.rand_felts(0xB1, 2))
    opk, ovk = ref.groth16_setup(r1, *tox)
    proof = ref.groth16_prove(opk, r1.n_public, a, b, c, wv, *rs)
    assert ref.groth16_verify(ovk, proof, wv[:r1.n_public])
    want = ref.groth16_proof_bytes(*proof).hex()
    pk_hex, vk_hex, h = fe.groth16_preprocess(rj, mont_limbs(list(tox)), keep_resident=True)
    assert pk_hex == pl.groth16_pk_bytes(opk).hex()
    assert vk_hex == pl.groth16_vk_bytes(dict(ovk, g1_beta=opk["g1_beta"], g1_delta=opk["g1_delta"])).hex()
    assert fe.groth16_prove_with_pk(rj, pk_hex, mont_limbs(list(rs))) == want
    assert fe.groth16_prove_with_pk(rj, None, mont_limbs(list(rs)), pk_handle=h) == want
    assert fe.groth16_prove_with_meta(rj, mont_limbs(list(tox)), mont_limbs(list(rs))) == want
    # drawn (r, s): another valid proof for the same key
    p2 = bytes.fromhex(fe.groth16_prove_with_pk(rj, pk_hex))
    assert p2.hex() != want
    pts = (pl.g1_decompress(p2[:32]), pl.g2_decompress(p2[32:96]), pl.g1_decompress(p2[96:]))
    assert ref.groth16_verify(ovk, pts, wv[:r1.n_public])
    # drawn toxic waste: keys of the same size, different bytes
    k2, v2 = fe.groth16_preprocess(rj)
    assert len(k2) == len(pk_hex) and len(v2) == len(vk_hex) and k2 != pk_hex
    zk.ProvingKey.from_handle(h, 0, 0, 0).free()
    for bad_pk in (pk_hex[:-2], "zz" + pk_hex[2:]):
        with pytest.raises(ValueError):
            fe.groth16_prove_with_pk(rj, bad_pk, mont_limbs(list(rs)))
    with pytest.raises(ValueError):
        fe.groth16_preprocess("{}")


def test_groth16_export_caches_change_no_byte_at_2p11_constraints():
    """What zk_groth16_prove_with_pk keeps resident (the circuit of a RawR1CS text minus its values string; the decoded key, window tables from its second
    proof on) against the path that keeps nothing: 2^10 gates = 2^11 constraints (tools/synth_raw_r1cs.py), pinned (r, s).  The first call (everything read,
    no tables), the second (tables built), a warm one, one after zk_export_cache_clear and the uncached chain zk_groth16_r1cs_from_raw (wire vector
    assembled on the HOST) -> zk_bn254_groth16_pk_read -> zk_bn254_groth16_prove_r1cs all give the same 128 bytes; another assignment of the same circuit
    reuses the resident circuit and gives another proof, which the host verifier accepts under its own public inputs only; the oracle's pairing verifier accepts
    the first proof under the key image Preprocess wrote; a text that differs OUTSIDE the values string is another circuit; another key of the same length is
    another key (the warm path's speculative proof with the resident one is dropped)."""
    from noir_backend_using_gnark_amd import frontend as fe
    from noir_backend_using_gnark_amd import verify as vf
    from tools import synth_raw_r1cs as sr
    fe.export_cache_clear()
    raw, w = sr.synth(1 << 10, 3, seed=0x51)
    raw2, w2 = sr.synth(1 << 10, 3, seed=0x51, first=(0xabcdef, 0x123456789))
    assert raw[:raw.index('"values"')] == raw2[:raw2.index('"values"')] and w != w2
    tox, rs = mont_limbs(list(ref.rand_felts(0xC0, 5))), mont_limbs(list(ref.rand_felts(0xC1, 2)))
    pk_hex, vk_hex = fe.groth16_preprocess(raw, tox)
    assert fe.export_cache_info()["circuits"] == 1 and fe.export_cache_info()["keys"] == 1  # the key Setup made is resident under the text it was written as
    fe.export_cache_clear()
    assert fe.export_cache_info() == {"circuits": 0, "keys": 0, "bytes": 0}
    first = fe.groth16_prove_with_pk(raw, pk_hex, rs)
    info = fe.export_cache_info()
    assert info["circuits"] == 1 and info["keys"] == 1
    second = fe.groth16_prove_with_pk(raw, pk_hex, rs)     # queues the key's window tables on the background thread (and proves without them)
    assert _lib.lib().zk_background_wait(C.c_int(-1)) == 1 and fe.export_cache_info()["bytes"] > info["bytes"]
    warm = fe.groth16_prove_with_pk(raw, pk_hex, rs)
    other = fe.groth16_prove_with_pk(raw2, pk_hex, rs)     # same circuit, other values: no second circuit
    assert fe.export_cache_info()["circuits"] == 1 and fe.export_cache_info()["keys"] == 1
    again = fe.groth16_prove_with_pk(raw, pk_hex, rs)
    fe.export_cache_clear()
    cleared = fe.groth16_prove_with_pk(raw, pk_hex, rs)
    assert first == second == warm == again == cleared and other != first
    # the chain that keeps nothing: host-assembled wires, a key read with its tables at once
    r1, d_w = fe.groth16_r1cs_from_raw(raw)
    rk = zk.ProvingKey.read_from(pk_hex)
    out = np.zeros(128, dtype=np.uint8)
    _lib.check(_lib.lib().zk_bn254_groth16_prove_r1cs(r1.handle, rk.handle, C.c_void_p(d_w.ptr), C.c_size_t(r1.n_wires), vp(rs[0]), vp(rs[1]), C.c_int(1), vp(out)))
    assert bytes(out).hex() == first
    rk.free(); r1.free(); d_w.free()
    # verification: public inputs read from the text against the resident circuit (host only)
    pub, pub2 = fe.groth16_public_inputs(raw), fe.groth16_public_inputs(raw2)
    assert pub.shape == (3, 4) and [pl.mont_np_to_ints(pub)[k] for k in range(3)] == w[:3]
    assert vf.groth16_verify(bytes.fromhex(first), vk_hex, pub) and vf.groth16_verify(bytes.fromhex(other), vk_hex, pub2)
    assert not vf.groth16_verify(bytes.fromhex(first), vk_hex, pub2)
    vkb = bytes.fromhex(vk_hex)
    nk = int.from_bytes(vkb[288:292], "big")
    ovk = dict(g1_alpha=pl.g1_decompress(vkb[0:32]), g2_beta=pl.g2_decompress(vkb[64:128]), g2_gamma=pl.g2_decompress(vkb[128:192]), g2_delta=pl.g2_decompress(vkb[224:288]),
               g1_ic=[pl.g1_decompress(vkb[292 + 32 * i:324 + 32 * i]) for i in range(nk)])
    pb = bytes.fromhex(first)
    assert ref.groth16_verify(ovk, (pl.g1_decompress(pb[:32]), pl.g2_decompress(pb[32:96]), pl.g1_decompress(pb[96:])), [1] + w[:3])
    # a text that differs outside the values string (one coefficient digit) is another circuit: read again, and the old key's proof for it does not verify
    at = raw.index('"constant_term":"') + len('"constant_term":"') + 63
    raw3 = raw[:at] + ("1" if raw[at] != "1" else "2") + raw[at + 1:]
    p3 = fe.groth16_prove_with_pk(raw3, pk_hex, rs)
    assert fe.export_cache_info()["circuits"] == 2 and p3 != first
    # another key of the SAME length for the first circuit while the first key is warm (tables built by the call above): the warm path proves with the resident
    # key of that length while the texts are still being compared, and must drop that proof -- what comes back is the second key's
    pk2_hex, vk2_hex = fe.groth16_preprocess(raw, mont_limbs(list(ref.rand_felts(0xC2, 5))))
    assert len(pk2_hex) == len(pk_hex) and pk2_hex != pk_hex
    q = bytes.fromhex(fe.groth16_prove_with_pk(raw, pk2_hex, rs))
    assert vf.groth16_verify(q, vk2_hex, pub) and not vf.groth16_verify(q, vk_hex, pub)
    assert fe.groth16_prove_with_pk(raw, pk_hex, rs) == first
    with pytest.raises(ValueError):
        fe.groth16_prove_with_pk(raw.replace(sr.felts_wire_hex(w)[8:72], "g" * 64), pk_hex, rs)  # a non-hex values string of the right length: the device decoder says so
    fe.export_cache_clear()


def test_groth16_setup_prove_verify_chain_at_2p15_constraints():
    """The whole Groth16 life cycle on the product side at a size the big-integer oracle cannot set up in test time: a satisfied random R1CS with 2^15
    constraints -> groth16.Setup on the device (pinned toxic waste) -> VerifyingKey.WriteTo / ProvingKey.WriteTo -> ReadFrom of the key image ->
    Prove from the witness -> the HOST verifier (pairings) accepts the 128 bytes under the written verifying key and rejects another public input, a
    proof with swapped points and the proof of another (r, s) against a key from other toxic waste.  A wrong Setup, Prove, key codec or verifier
    breaks the pairing equation."""
    from noir_backend_using_gnark_amd import verify as zv
    g = ref.SplitMix64(0x515)
    nc, n_in, npub = 1 << 15, 500, 4                      # wires: [ONE, 3 public, 497 secret inputs, 2^15 products]
    w = [1] + [g.felt() for _ in range(n_in - 1)]
    one = mont_limbs([1])[0]
    coeff_cache = {}

    def co(v):
        if v not in coeff_cache:
            coeff_cache[v] = mont_limbs([v])[0]
        return coeff_cache[v]

    cons = []
    for j in range(nc):
        la, lb, ra, rb = (int(g.next() % n_in) for _ in range(4))
        c1, c2 = 1 + int(g.next() % 7), 1 + int(g.next() % 7)
        L = {la: co(c1)}
        L[lb] = co((c2 + (c1 if lb == la else 0)) % ref.R)
        Rr = {ra: one}
        if rb != ra:
            Rr[rb] = one
        lv = (c1 * w[la] + c2 * w[lb]) % ref.R
        rv = (w[ra] + (w[rb] if rb != ra else 0)) % ref.R
        cons.append((L, Rr, {n_in + j: one}))
        w.append(lv * rv % ref.R)
    r1 = zk.R1CS(npub, len(w), cons)
    wm = mont_limbs(w)
    tox = mont_limbs(ref.rand_felts(0x70C5, 5))
    pk, vk = zk.setup(r1, tox)
    vkb = pk.vk_write_to(vk)
    img = pk.write_to()
    pk.free()
    rk = zk.ProvingKey.read_from(img)
    r, s = mont_limbs(ref.rand_felts(0x70C6, 2))
    proof = zk.prove_r1cs(r1, rk, wm, r, s)
    pub = wm[1:npub]
    assert zv.groth16_verify(proof, vkb, pub)
    bad = pub.copy()
    bad[0] = mont_limbs([(w[1] + 1) % ref.R])[0]
    assert not zv.groth16_verify(proof, vkb, bad)
    assert not zv.groth16_verify(proof[96:] + proof[32:96] + proof[:32], vkb, pub)      # Ar and Krs swapped
    pk2, vk2 = zk.setup(r1, mont_limbs(ref.rand_felts(0x70C7, 5)))
    assert not zv.groth16_verify(proof, pk2.vk_write_to(vk2), pub)                       # a key from other toxic waste
    p2 = zk.prove_r1cs(r1, pk2, wm, r, s)
    assert zv.groth16_verify(p2, pk2.vk_write_to(vk2), pub) and p2 != proof
    for k in (rk, pk2):
        k.free()
    r1.free()


def test_key_and_srs_readers_survive_mutated_images_on_the_device():
    """Mutation fuzz with the device in the loop: random byte flips / truncations / insertions of valid Groth16 key, PLONK key and SRS images go through
    zk_bn254_groth16_pk_read, zk_bn254_plonk_pk_read and zk_bn254_kzg_srs_read -- every call either loads (and the key then still serialises and frees)
    or comes back as ValueError; afterwards the untouched images still load and the library still proves (no poisoned state, no leaked stream slot)."""
    import random
    from noir_backend_using_gnark_amd import kzg, plonk as zp
    rnd = random.Random(0xD1CE)
    wire = json.load(open(os.path.join(HERE, "golden", "groth16_wire_golden.json")))[1]
    good = bytes.fromhex(wire["pk_hex"])

    def mutate(b):
        b = bytearray(b)
        for _ in range(rnd.randint(1, 3)):
            p = rnd.randrange(len(b))
            kind = rnd.randint(0, 3)
            if kind == 0:
                b[p] ^= 1 << rnd.randrange(8)
            elif kind == 1:
                b[p] = rnd.randrange(256)
            elif kind == 2:
                b = b[:p] if p > 8 else b
            else:
                b[p:p] = bytes([rnd.randrange(256)])
        return bytes(b)

    loaded = 0
    for it in range(300):
        m = mutate(good)
        try:
            k = zk.ProvingKey.read_from(m.hex() if it & 1 else m, is_hex=bool(it & 1), precompute_tables=False)
        except ValueError:
            continue
        loaded += 1          # e.g. a flipped InfinityA/B pair that stays consistent, or a sign flag: another valid key
        assert len(k.write_to()) > 0
        k.free()
    assert loaded < 300
    # SRS images
    srs = kzg.new_srs(40, mont_limbs([0x1234567])[0])
    img = srs.write()
    srs.free()
    for it in range(200):
        try:
            s2 = kzg.read_srs(mutate(img), is_hex=False, table_window_bits=-1)
        except ValueError:
            continue
        s2.free()
    kzg.read_srs(img, is_hex=False).free()
    # the library is intact: the untouched key loads and proves the golden proof
    g = [e for e in json.load(open(os.path.join(HERE, "golden", "bn254_golden.json")))["groth16"] if e["name"] == "seq_r1cs_13"][0]
    rk = zk.ProvingKey.read_from(good)
    a, b, c, w = (mont_limbs([h2i(v) for v in g[k]]) for k in ("a", "b", "c", "w"))
    assert zk.prove(rk, a, b, c, w, mont_limbs([h2i(g["r"])])[0], mont_limbs([h2i(g["s"])])[0]).hex() == g["proof"]
    rk.free()
