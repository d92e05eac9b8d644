"""CPU suite for the PLONK restatement (oracle/plonk_ref.py): the committed fixtures -- the reference's three demo circuits
(gnark_backend_ffi/main.go:223-248, lowered per backend/plonk/sparse_r1cs.go:44-107) -- reproduce byte for byte, verify by pairings,
fail when tampered with; the transcript / KZG pieces satisfy their defining identities; the C-oracle-accelerated path agrees with
the pure-Python one."""
import json
import os

import pytest

from oracle import bn254_ref as ref
from oracle import plonk_ref as pl

R = ref.R
HERE = os.path.dirname(os.path.abspath(__file__))
h2i = lambda h: int(h, 16)


@pytest.fixture(scope="module")
def plonk_golden():
    return json.load(open(os.path.join(HERE, "golden", "plonk_golden.json")))


def _instance(e):
    spr, sol = pl.sparse_r1cs_from_acir(e["acir"], [h2i(v) for v in e["values"]])
    srs = pl.kzg_new_srs(e["srs_size"], h2i(e["srs_alpha"]))
    return spr, sol, srs


def test_lowering_matches_fixture_and_is_satisfied(plonk_golden):
    for e in plonk_golden:
        spr, sol, _ = _instance(e)
        assert spr.n_public == e["n_public"] and spr.n_vars == e["n_vars"]
        assert [[("%064x" % c) for c in g[:5]] + list(g[5:]) for g in spr.constraints] == e["gates"]
        assert ["%064x" % v for v in sol] == e["solution"]
        assert spr.is_satisfied(sol)
    # the reference's "-1" literal (main.go:233) is r - 1
    assert h2i(plonk_golden[0]["gates"][0][1]) == R - 1


def test_golden_proofs_reproduce_and_verify(plonk_golden):
    e = plonk_golden[0]
    spr, sol, srs = _instance(e)
    pk, vk = pl.plonk_setup(spr, srs)
    assert {k: ref.g1_affine_mont_bytes(vk[k]).hex() for k in ("ql", "qr", "qm", "qo", "qk")} == {k: e["vk"][k] for k in ("ql", "qr", "qm", "qo", "qk")}
    assert [ref.g1_affine_mont_bytes(p).hex() for p in vk["s"]] == e["vk"]["s"]
    assert pl.plonk_vk_bytes(vk).hex() == e["vk_hex"] and pl.plonk_pk_bytes(pk).hex() == e["pk_hex"] and e["pk_hex"].startswith(e["vk_hex"])
    back = pl.plonk_pk_from_bytes(bytes.fromhex(e["pk_hex"]))
    assert back["perm"] == list(pk["perm"]) and back["ql"] == list(pk["ql"]) and back["s3"] == list(pk["s3"])
    trace = {}
    proof = pl.plonk_prove(pk, sol, [h2i(v) for v in e["blinders"]], trace=trace)
    assert pl.plonk_proof_bytes(proof).hex() == e["proof"] and len(e["proof"]) == 2 * 548
    assert {k: "%064x" % trace[k] for k in e["challenges"]} == e["challenges"]
    pub = sol[:spr.n_public]
    assert pl.plonk_verify(vk, proof, pub)
    assert not pl.plonk_verify(vk, proof, [(pub[0] + 1) % R])                                  # another public input
    assert not pl.plonk_verify(vk, dict(proof, claimed=[(proof["claimed"][0] + 1) % R] + proof["claimed"][1:]), pub)
    assert not pl.plonk_verify(vk, dict(proof, z=ref.g1_add(proof["z"], ref.G1_GEN)), pub)    # another commitment
    # pinned challenges: same entry point, other bytes, still a valid proof under the same pinned values
    pinned = {k: h2i(v) for k, v in e["pinned_challenges"].items()}
    proof_p = pl.plonk_prove(pk, sol, [h2i(v) for v in e["blinders"]], challenges=pinned)
    assert pl.plonk_proof_bytes(proof_p).hex() == e["proof_pinned"] != e["proof"]


def test_unsatisfied_witness_is_rejected(plonk_golden):
    e = plonk_golden[1]
    spr, sol, srs = _instance(e)
    pk, _ = pl.plonk_setup(spr, srs)
    bad = list(sol)
    bad[-2] = (bad[-2] + 1) % R   # w5 (the last variable, w6, is in no gate) enters three gates
    assert not spr.is_satisfied(bad)
    with pytest.raises(AssertionError, match="not satisfied"):
        pl.plonk_prove(pk, bad, [1] * 9)


def test_transcript_and_kzg_identities():
    import hashlib
    t = pl.Transcript("gamma", "beta")
    t.bind("gamma", b"\x01\x02")
    g = t.compute("gamma")
    assert g == hashlib.sha256(b"gamma" + b"\x01\x02").digest()
    assert t.compute("beta") == hashlib.sha256(b"beta" + g).digest()
    # synthetic division: f = q * (X - a) + f(a)
    f = ref.rand_felts(5, 9)
    a = 12345
    fa = pl.poly_eval(f, a)
    q = pl.divide_by_x_minus_a(f, fa, a)
    x = 777
    assert (pl.poly_eval(q, x) * (x - a) + fa) % R == pl.poly_eval(f, x)
    # an opening verifies by pairings
    srs = pl.kzg_new_srs(9, 424242)
    be = pl._Backend(False)
    c, h = be.msm_g1(srs["g1"], f), be.msm_g1(srs["g1"], q)
    assert pl._kzg_check(c, h, fa, a, srs["g2"]) and not pl._kzg_check(c, h, (fa + 1) % R, a, srs["g2"])


def test_fast_backend_equals_pure_python():
    """2^7-row random satisfiable circuit: NTTs / MSMs through the C oracle give the pure-Python proof bytes."""
    g = ref.SplitMix64(0x77)
    nvars, nc, npub = 40, 100, 3
    sol = [g.felt() for _ in range(nvars)]
    gates = []
    for _ in range(nc):
        xa, xb, xc = (int(g.next() % nvars) for _ in range(3))
        ql, qr, qo, qm = (g.felt() for _ in range(4))
        qk = (-(ql * sol[xa] + qr * sol[xb] + qo * sol[xc] + qm * sol[xa] * sol[xb])) % R
        gates.append((ql, qr, qo, qm, qk, xa, xb, xc))
    spr = pl.SparseR1CS(npub, nvars - npub, gates)
    assert spr.is_satisfied(sol)
    srs = pl.kzg_new_srs(128 + 3, 99991, fast=True)
    srs_py = dict(g1=[pl.g1_from_np(p) for p in srs["g1"]], g2=srs["g2"])
    bl = ref.rand_felts(1, 9)
    pk_f, vk_f = pl.plonk_setup(spr, srs, fast=True)
    pk_p, vk_p = pl.plonk_setup(spr, srs_py, fast=False)
    assert vk_f["s"] == vk_p["s"] and vk_f["qk"] == vk_p["qk"]
    pf = pl.plonk_prove(pk_f, sol, bl, fast=True)
    pp = pl.plonk_prove(pk_p, sol, bl, fast=False)
    assert pl.plonk_proof_bytes(pf) == pl.plonk_proof_bytes(pp)
    assert pl.plonk_verify(vk_f, pf, sol[:npub])


# ------------------------------------------------------------------------------------------------ the C / OpenMP twin (oracle/plonk_oracle_impl.h)
def _c_key(spr, srs_np, nthreads=0):
    from oracle import oracle as orc
    co = [pl.ints_to_mont_np([g[k] for g in spr.constraints]) if spr.constraints else pl._np().zeros((0, 4), dtype="uint64") for k in (0, 1, 3, 2, 4)]  # ql, qr, qm, qo, qk
    xs = [[g[k] for g in spr.constraints] for k in (5, 6, 7)]
    return orc.PlonkKeyC(spr.n_public, spr.n_vars, *co, *xs, srs_np, nthreads=nthreads)


def test_c_sha256_is_sha256():
    import hashlib
    from oracle import oracle as orc
    for n in (0, 1, 55, 56, 63, 64, 65, 119, 120, 1000):
        data = bytes((7 * i + n) & 0xFF for i in range(n))
        assert orc.sha256(data) == hashlib.sha256(data).digest(), n


def test_c_plonk_oracle_reproduces_the_fixtures(plonk_golden):
    """orc_plonk_setup / orc_plonk_prove on the reference's three demo circuits: the key's polynomials, the verifying key's digests, the permutation,
    the five Fiat-Shamir challenges and the 548 proof bytes equal the pure-Python restatement's and the committed fixtures."""
    np = pl._np()
    for e in plonk_golden:
        spr, sol, srs = _instance(e)
        srs_np = np.stack([pl.g1_to_np(p) for p in srs["g1"]])
        pk, vk = pl.plonk_setup(spr, srs)
        ck = _c_key(spr, srs_np)
        assert ck.n == pk["n"] and ck.n4 == pk["d1"].n
        for name in ck.NAMES:
            assert pl.mont_np_to_ints(ck.poly(name)) == list(pk[name]), (e["name"], name)
        assert list(ck.perm()) == list(pk["perm"])
        dig = ck.vk_digests()
        want = [vk["s"][0], vk["s"][1], vk["s"][2], vk["ql"], vk["qr"], vk["qm"], vk["qo"], vk["qk"]]
        assert [pl.g1_from_np(d) for d in dig] == want
        proof, ch = ck.prove(pl.ints_to_mont_np(sol), pl.ints_to_mont_np([h2i(v) for v in e["blinders"]]), want_challenges=True)
        assert proof.hex() == e["proof"], e["name"]
        assert {k: "%064x" % ch[k] for k in e["challenges"]} == e["challenges"]
        ck.free()


def test_c_plonk_oracle_equals_python_on_a_random_circuit_and_refuses_a_bad_witness():
    """2^9 rows, random wiring with repeated variables (long permutation cycles), several public inputs, an SRS of exactly n + 3 points; one thread and many
    give the same bytes; a witness that violates a gate is refused like the Python prover refuses it."""
    import random
    np = pl._np()
    rng = random.Random(0xC0FFEE)
    npub, nvars, nc = 3, 200, 505
    sol = [rng.randrange(R) for _ in range(nvars)]
    gates = []
    for _ in range(nc):
        xa, xb, xc = (rng.randrange(nvars) for _ in range(3))
        ql, qr, qo, qm = (rng.randrange(R) for _ in range(4))
        qc = (-(ql * sol[xa] + qr * sol[xb] + qo * sol[xc] + qm * sol[xa] * sol[xb])) % R
        gates.append((ql, qr, qo, qm, qc, xa, xb, xc))
    spr = pl.SparseR1CS(npub, nvars - npub, gates)
    assert spr.is_satisfied(sol)
    srs = pl.kzg_new_srs(512 + 3, 0x1234567, fast=True)
    bl = [rng.randrange(R) for _ in range(9)]
    pk, vk = pl.plonk_setup(spr, srs, fast=True)
    want = pl.plonk_proof_bytes(pl.plonk_prove(pk, sol, bl, fast=True))
    assert pl.plonk_verify(vk, pl.plonk_prove(pk, sol, bl, fast=True), sol[:npub])
    srs_np = np.ascontiguousarray(srs["g1"])
    for nthreads in (1, 0):
        ck = _c_key(spr, srs_np, nthreads)
        assert ck.prove(pl.ints_to_mont_np(sol), pl.ints_to_mont_np(bl)) == want
        bad = list(sol)
        bad[gates[7][5]] = (bad[gates[7][5]] + 1) % R
        with pytest.raises(AssertionError, match="not satisfied"):
            ck.prove(pl.ints_to_mont_np(bad), pl.ints_to_mont_np(bl))
        ck.free()
    with pytest.raises(ValueError):
        _c_key(spr, srs_np[:512 + 2])  # an SRS one point short


def test_c_oracle_under_address_and_ub_sanitizers(tmp_path):
    """oracle/build/libbn254_oracle_asan.so (-fsanitize=address,undefined; the GPU side has no sanitizer on this pool, the checker does): PLONK Setup + Prove on circuits
    around the domain edges (sizeSystem < 6 -> 8x domain; one public input; no public input; 1,500 gates) and the transform on both sides of the size from which the cache-tiled
    schedule runs (2^17), forward . inverse = identity in four modes -- no report."""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "build/libbn254_oracle_asan.so"], stdout=subprocess.DEVNULL)
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    code = r"""
import sys, random
sys.path.insert(0, %r)
from oracle import oracle as orc
orc._SO = %r
orc.build = lambda force=False: orc._SO
from oracle import plonk_ref as pl, bn254_ref as ref
R = ref.R
rng = random.Random(7)
for (npub, nvars, nc) in ((0, 5, 3), (1, 4, 5), (3, 60, 61), (2, 700, 1500)):
    sol = [rng.randrange(R) for _ in range(nvars)]
    gates = []
    for _ in range(nc):
        xa, xb, xc = (rng.randrange(nvars) for _ in range(3))
        ql, qr, qo, qm = (rng.randrange(R) for _ in range(4))
        gates.append((ql, qr, qo, qm, (-(ql * sol[xa] + qr * sol[xb] + qo * sol[xc] + qm * sol[xa] * sol[xb])) %% R, xa, xb, xc))
    n = 1
    while n < nc + npub: n <<= 1
    ck = orc.PlonkKeyC(npub, nvars, *[pl.ints_to_mont_np([g[k] for g in gates]) for k in (0, 1, 3, 2, 4)], *[[g[k] for g in gates] for k in (5, 6, 7)], orc.g1_gen_points(5, n + 3))
    assert len(ck.prove(pl.ints_to_mont_np(sol), pl.ints_to_mont_np([rng.randrange(R) for _ in range(9)]))) == 548
    ck.free()
for logn in (3, 12, 16, 17, 18):
    x = orc.rand_fr(logn, 1 << logn)
    assert (orc.fr_ntt(orc.fr_ntt(x, False, orc.DIF), True, orc.DIT) == x).all()
    assert (orc.fr_ntt(orc.fr_ntt(x, False, orc.DIT, True), True, orc.DIF, True) == x).all()
print("sanitized run ok")
""" % (root, os.path.join(root, "oracle", "build", "libbn254_oracle_asan.so"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0"))
    assert out.returncode == 0 and "sanitized run ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
