"""GPU parity suite (-m gpu): the HIP path, called through the C ABI (libzkmi.so via ctypes), against the CPU oracle on the
same seeded inputs, against the committed golden vectors, and -- at BASELINE sizes -- through size-independent properties.
Bit-exact everywhere: this path is integer arithmetic only."""
import ctypes as C

import numpy as np
import pytest

import noir_backend_using_gnark_amd as zk
from noir_backend_using_gnark_amd import _lib
from noir_backend_using_gnark_amd import bn254 as zb
from oracle import bn254_ref as ref
from oracle import oracle as orc
from tests.helpers import (from_mont_limbs, g1_points_from_scalars, g2_points_from_scalars, golden_pk, h2i, mont_limbs, sha_image)

pytestmark = pytest.mark.gpu

# the test vectors are Montgomery fr.Element images; MultiExpConfig{} (upstream's zero value) means regular-form scalars
MONT = zk.MultiExpConfig(scalars_mont=True)


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    _lib.require_device()  # fail loudly: no silent fallback


# ------------------------------------------------------------------------------------------------ field / generators
def test_fr_mul_on_device_matches_oracle():
    n = 4096
    a, b = orc.rand_fr(101, n), orc.rand_fr(102, n)
    edge = mont_limbs([0, 1, ref.R - 1, ref.R - 2, 2, (ref.R - 1) // 2, ref.MONT_R % ref.R, 3])
    a[:8] = edge
    b[:8] = edge[::-1]
    da, db, do = _lib.DeviceBuffer.from_numpy(a), _lib.DeviceBuffer.from_numpy(b), _lib.DeviceBuffer(n * 32)
    _lib.check(_lib.lib().zk_bn254_fr_mul_dev(C.c_void_p(do.ptr), C.c_void_p(da.ptr), C.c_void_p(db.ptr), C.c_size_t(n), None))
    got = do.to_numpy(np.uint64, (n, 4))
    exp = np.stack([orc.fe_op("mul", 0, a[i], b[i]) for i in range(n)])
    assert (got == exp).all()


@pytest.mark.parametrize("mont", [0, 1])
@pytest.mark.parametrize("witness", [0, 1])
def test_fr_random_matches_oracle(mont, witness):
    n = 3000
    d = _lib.DeviceBuffer(n * 32)
    _lib.check(_lib.lib().zk_bn254_fr_random_dev(C.c_void_p(d.ptr), C.c_size_t(n), C.c_uint64(0xC0FFEE), C.c_int(mont), C.c_int(witness), None))
    assert (d.to_numpy(np.uint64, (n, 4)) == orc.rand_fr(0xC0FFEE, n, mont=bool(mont), witness_like=bool(witness))).all()


def test_generate_points_match_oracle():
    n = 96
    d1, d2 = _lib.DeviceBuffer(n * 64), _lib.DeviceBuffer(n * 128)
    _lib.check(_lib.lib().zk_bn254_g1_generate_dev(C.c_void_p(d1.ptr), C.c_size_t(n), C.c_uint64(77), None))
    _lib.check(_lib.lib().zk_bn254_g2_generate_dev(C.c_void_p(d2.ptr), C.c_size_t(n), C.c_uint64(77), None))
    p1, p2 = d1.to_numpy(np.uint64, (n, 8)), d2.to_numpy(np.uint64, (n, 16))
    assert (p1 == orc.g1_gen_points(77, n)).all()
    assert (p2 == orc.g2_gen_points(77, n)).all()
    assert all(orc.g1_on_curve(p) for p in p1[:8]) and all(orc.g2_on_curve(p) for p in p2[:8])


# ------------------------------------------------------------------------------------------------ NTT
def test_ntt_golden_modes(golden):
    for e in golden["ntt"]:
        if e["kind"] == "survey_ntt4":
            x = mont_limbs([h2i(v) for v in e["input"]])
            zk.Domain(4).fft(x, zk.DIF)
            assert from_mont_limbs(x) == [h2i(v) for v in e["dif"]]
            zk.bit_reverse(x)
            assert from_mont_limbs(x) == [h2i(v) for v in e["natural"]]
            continue
        n = 1 << e["log_n"]
        x = mont_limbs(ref.rand_felts(e["seed"], n))
        d = zk.Domain(n)
        (d.fft_inverse if e["inverse"] else d.fft)(x, e["decimation"], bool(e["coset"]))
        assert sha_image(x) == e["sha256"], e
        if "output" in e:
            assert from_mont_limbs(x) == [h2i(v) for v in e["output"]]


@pytest.mark.parametrize("log_n", [11, 12, 14, 16, 19, 20])
def test_ntt_vs_oracle_all_modes(log_n):
    n = 1 << log_n
    x = orc.rand_fr(500 + log_n, n)
    modes = [(i, d, c) for i in (0, 1) for d in (zk.DIT, zk.DIF) for c in (0, 1)]
    if log_n >= 19:
        modes = [(0, zk.DIF, 0), (1, zk.DIF, 0), (0, zk.DIT, 1), (1, zk.DIF, 1)]  # the four combinations computeH uses
    dom = zk.Domain(n)
    for inverse, dec, coset in modes:
        y = x.copy()
        (dom.fft_inverse if inverse else dom.fft)(y, dec, bool(coset))
        exp = orc.fr_ntt(x, bool(inverse), dec, bool(coset))
        assert (y == exp).all(), (log_n, inverse, dec, coset)


def test_ntt_roundtrip_full_size_on_device():
    """2^22: FFTInverse(DIT) . FFT(DIF) == identity, data resident in HBM (size-independent property)."""
    log_n = 22
    n = 1 << log_n
    d = _lib.DeviceBuffer(n * 32)
    _lib.check(_lib.lib().zk_bn254_fr_random_dev(C.c_void_p(d.ptr), C.c_size_t(n), C.c_uint64(9), C.c_int(1), C.c_int(0), None))
    x0 = d.to_numpy(np.uint64, (n, 4))
    dom = zk.Domain(n)
    for coset in (False, True):
        dom.fft(d, zk.DIF, coset)
        mid = d.to_numpy(np.uint64, (n, 4))
        assert not (mid == x0).all()
        dom.fft_inverse(d, zk.DIT, coset)
        assert (d.to_numpy(np.uint64, (n, 4)) == x0).all()
    # spot check against the oracle on the same input (transform of size 2^22 takes the oracle a few seconds)
    dom.fft(d, zk.DIF)
    assert sha_image(d.to_numpy(np.uint64, (n, 4))) == sha_image(orc.fr_ntt(x0, False, orc.DIF))


def test_bit_reverse():
    for log_n in (0, 1, 5, 13):
        x = orc.rand_fr(3, 1 << log_n)
        y = x.copy()
        zk.bit_reverse(y)
        assert (y == orc.fr_bit_reverse(x)).all()


def test_ntt_argument_errors():
    with pytest.raises(ValueError):
        zk.Domain(8).fft(np.zeros((4, 4), np.uint64), zk.DIF)
    with pytest.raises(ValueError):
        zk.Domain(8).fft(np.zeros((8, 4), np.uint64), 7)
    with pytest.raises(ValueError):
        zk.Domain(1 << 29)


# ------------------------------------------------------------------------------------------------ MSM
@pytest.mark.parametrize("c", [0, 4, 9, 16])
def test_msm_golden(golden, c):
    for e in golden["msm"]:
        pts1 = g1_points_from_scalars([h2i(x) for x in e["point_scalars"]])
        pts2 = g2_points_from_scalars([h2i(x) for x in e["point_scalars"]])
        sc = [h2i(x) for x in e["scalars"]]
        cfg = zk.MultiExpConfig(scalars_mont=True, window_bits=c)
        assert zk.g1_multi_exp(pts1, mont_limbs(sc), cfg).tobytes().hex() == e["g1"], (e["kind"], c)
        assert zk.g2_multi_exp(pts2, mont_limbs(sc), cfg).tobytes().hex() == e["g2"], (e["kind"], c)
        cfg2 = zk.MultiExpConfig(window_bits=c)  # upstream default: regular-form scalars
        assert zk.g1_multi_exp(pts1, orc.ints_to_limbs(sc), cfg2).tobytes().hex() == e["g1"]


def test_msm_errors_and_empty():
    p, s = np.zeros((3, 8), np.uint64), np.zeros((2, 4), np.uint64)
    with pytest.raises(ValueError, match=r"len\(points\) != len\(scalars\)"):
        zk.g1_multi_exp(p, s, config=MONT)
    with pytest.raises(ValueError, match="NbTasks"):
        zk.g1_multi_exp(p, np.zeros((3, 4), np.uint64), zk.MultiExpConfig(scalars_mont=True, nb_tasks=2000))
    assert (zk.g1_multi_exp(np.zeros((0, 8), np.uint64), np.zeros((0, 4), np.uint64), config=MONT) == 0).all()
    assert (zk.g2_multi_exp(np.zeros((0, 16), np.uint64), np.zeros((0, 4), np.uint64), config=MONT) == 0).all()
    # all-zero scalars, all-infinity points
    pts = orc.g1_gen_points(1, 10)
    assert (zk.g1_multi_exp(pts, np.zeros((10, 4), np.uint64), config=MONT) == 0).all()
    assert (zk.g1_multi_exp(np.zeros((10, 8), np.uint64), orc.rand_fr(2, 10), config=MONT) == 0).all()


@pytest.mark.parametrize("n,seed", [(1, 1), (2, 2), (63, 3), (1000, 4), (4097, 5), (1 << 16, 6)])
def test_g1_msm_vs_oracle_uniform(n, seed):
    pts, sc = orc.g1_gen_points(seed, n), orc.rand_fr(seed + 100, n)
    assert (zk.g1_multi_exp(pts, sc, config=MONT) == orc.g1_msm(pts, sc)).all()


@pytest.mark.parametrize("n,c", [(20000, 0), (20000, 16), (3000, 5)])
def test_g1_msm_witness_like_heavy_buckets(n, c):
    """50% of the scalars in {0,1}: one bucket holds a quarter of all points (exercises task splitting + fold)."""
    pts, sc = orc.g1_gen_points(11, n), orc.rand_fr(12, n, witness_like=True)
    assert (zk.g1_multi_exp(pts, sc, zk.MultiExpConfig(scalars_mont=True, window_bits=c)) == orc.g1_msm(pts, sc)).all()


def test_g1_msm_collisions_and_cancellation():
    """Repeated points (P+P -> doubling inside a bucket), P + (-P) -> infinity inside a bucket, all scalars equal."""
    n = 512
    base = orc.g1_gen_points(21, 4)
    pts = np.tile(base, (n // 4, 1))
    sc = np.tile(orc.rand_fr(22, 1), (n, 1))
    assert (zk.g1_multi_exp(pts, sc, zk.MultiExpConfig(scalars_mont=True, window_bits=8)) == orc.g1_msm(pts, sc)).all()
    # scalars s and r - s on the same point cancel
    s_int = ref.rand_felts(23, n // 2)
    sc2 = mont_limbs(s_int + [(ref.R - v) % ref.R for v in s_int])
    pts2 = np.concatenate([orc.g1_gen_points(24, n // 2)] * 2)
    assert (zk.g1_multi_exp(pts2, sc2, config=MONT) == 0).all()
    assert (orc.g1_msm(pts2, sc2) == 0).all()


@pytest.mark.parametrize("kind", ["all_ones", "sixty_values", "hundreds_of_values", "half_ones_half_uniform", "one_and_minus_one"])
def test_msm_giant_buckets_plain_and_table_paths(kind):
    """Extremely skewed scalars -- what makes buckets of tens of thousands of points: the device then shortens the tasks and folds the giant buckets'
    partial sums with many waves over segments (k_pick_len, k_fold_giant); more giants than the list holds fall back to the serial fold.  Plain MSM and
    the window-table path (one bucket set for all windows), G1 at 2^17 and G2 at 2^14 points, against the oracle."""
    n = 1 << 17
    g = ref.SplitMix64(0x61A7)
    if kind == "all_ones":
        vals = [1] * n
    elif kind == "sixty_values":            # 60 giant buckets in the lowest window: more than the giant list's 48 entries
        vals = [1 + int(g.next() % 60) for _ in range(n)]
    elif kind == "hundreds_of_values":      # ~400 buckets of ~330 points: split buckets, no giants
        vals = [1 + int(g.next() % 400) for _ in range(n)]
    elif kind == "half_ones_half_uniform":
        vals = [1 if i & 1 else g.felt() for i in range(n)]
    else:                                   # 1 and r - 1: two giant buckets with opposite signs in every window of the signed recoding
        vals = [1 if g.next() & 1 else ref.R - 1 for _ in range(n)]
    sc = mont_limbs(vals)
    pts = orc.g1_gen_points(0x900 + len(kind), n)
    want = orc.g1_msm(pts, sc)
    assert (zk.g1_multi_exp(pts, sc, config=MONT) == want).all(), kind
    rb = zb.ResidentBases(pts)              # window tables: every window feeds one bucket set
    assert (rb.multi_exp(sc, config=MONT) == want).all(), kind
    rb.free()
    m = 1 << 14
    p2 = orc.g2_gen_points(0x910 + len(kind), m)
    assert (zk.g2_multi_exp(p2, sc[:m], config=MONT) == orc.g2_msm(p2, sc[:m])).all(), kind


@pytest.mark.parametrize("n,witness", [(1, False), (500, False), (5000, True), (1 << 14, False)])
def test_g2_msm_vs_oracle(n, witness):
    pts, sc = orc.g2_gen_points(31, n), orc.rand_fr(32, n, witness_like=witness)
    assert (zk.g2_multi_exp(pts, sc, config=MONT) == orc.g2_msm(pts, sc)).all()


def test_msm_resident_bases_and_device_pointers():
    n = 5000
    pts, sc = orc.g1_gen_points(41, n), orc.rand_fr(42, n)
    exp = orc.g1_msm(pts, sc)
    rb = zb.ResidentBases(pts)
    assert (rb.multi_exp(sc, config=MONT) == exp).all()
    assert (rb.multi_exp(sc[100:600], offset=100, config=MONT) == orc.g1_msm(pts[100:600], sc[100:600])).all()
    with pytest.raises(ValueError):
        rb.multi_exp(sc, offset=1, config=MONT)
    rb.free()
    dp, ds = _lib.DeviceBuffer.from_numpy(pts), _lib.DeviceBuffer.from_numpy(sc)
    assert (zb.g1_multi_exp_dev(dp.ptr, ds.ptr, n, config=MONT) == exp).all()
    # range-sharded partials (what two ranks would compute) combine to the same point
    h = n // 2
    parts = np.stack([zb.g1_multi_exp_dev(dp.ptr, ds.ptr, h, partial=True, config=MONT),
                      zb.g1_multi_exp_dev(dp.ptr + h * 64, ds.ptr + h * 32, n - h, partial=True, config=MONT)])
    assert (zb.g1_sum_partials(parts) == exp).all()
    p2 = orc.g2_gen_points(43, 300)
    d2, s2 = _lib.DeviceBuffer.from_numpy(p2), _lib.DeviceBuffer.from_numpy(sc[:300])
    parts2 = np.stack([zb.g2_multi_exp_dev(d2.ptr, s2.ptr, 100, partial=True, config=MONT), zb.g2_multi_exp_dev(d2.ptr + 100 * 128, s2.ptr + 100 * 32, 200, partial=True, config=MONT)])
    assert (zb.g2_sum_partials(parts2) == orc.g2_msm(p2, sc[:300])).all()


def test_g1_msm_full_size_properties():
    """2^20 points generated on the device: (1) result equals the multi-threaded oracle's; (2) homogeneity
    MSM(P, k*s) == k * MSM(P, s) with the scalars scaled on the device."""
    n = 1 << 20
    L = _lib.lib()
    dp, ds, dk, dks = (_lib.DeviceBuffer(n * 64), _lib.DeviceBuffer(n * 32), _lib.DeviceBuffer(n * 32), _lib.DeviceBuffer(n * 32))
    _lib.check(L.zk_bn254_g1_generate_dev(C.c_void_p(dp.ptr), C.c_size_t(n), C.c_uint64(0xB1), None))
    _lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(ds.ptr), C.c_size_t(n), C.c_uint64(0xC), C.c_int(1), C.c_int(0), None))
    k = orc.rand_fr(0xD, 1)
    dk_host = np.tile(k, (n, 1))
    _lib.check(L.zk_dev_h2d(C.c_void_p(dk.ptr), dk_host.ctypes.data_as(C.c_void_p), C.c_size_t(n * 32)))
    _lib.check(L.zk_bn254_fr_mul_dev(C.c_void_p(dks.ptr), C.c_void_p(ds.ptr), C.c_void_p(dk.ptr), C.c_size_t(n), None))
    r1 = zb.g1_multi_exp_dev(dp.ptr, ds.ptr, n, config=MONT)
    r2 = zb.g1_multi_exp_dev(dp.ptr, dks.ptr, n, config=MONT)
    assert orc.g1_on_curve(r1)
    assert (orc.g1_mul(r1, k[0]) == r2).all()
    pts, sc = dp.to_numpy(np.uint64, (n, 8)), ds.to_numpy(np.uint64, (n, 4))
    assert (orc.g1_msm(pts, sc) == r1).all()
    # witness-like scalars at full size
    _lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(ds.ptr), C.c_size_t(n), C.c_uint64(0xC), C.c_int(1), C.c_int(1), None))
    assert (zb.g1_multi_exp_dev(dp.ptr, ds.ptr, n, config=MONT) == orc.g1_msm(pts, ds.to_numpy(np.uint64, (n, 4)))).all()


# ------------------------------------------------------------------------------------------------ Groth16
def test_compute_h_golden_and_oracle(golden):
    for e in golden["compute_h"]:
        a, b, c = (mont_limbs(ref.rand_felts(s, e["n"])) for s in e["seeds"])
        assert sha_image(zk.compute_h(a, b, c, e["log_n"])) == e["sha256"]
    for log_n, n in ((12, 4000), (16, 1 << 16)):
        a, b, c = orc.rand_fr(1, n), orc.rand_fr(2, n), orc.rand_fr(3, n)
        assert (zk.compute_h(a, b, c, log_n) == orc.groth16_compute_h(a, b, c, log_n)).all()


@pytest.mark.parametrize("log_n,n", [(1, 2), (2, 3), (11, 2048), (21, (1 << 21) - 3)])
def test_compute_h_pass_counts_vs_oracle(log_n, n):
    """computeH where its shared launches change shape: a single contiguous pass (<= 2^11: the closing transform's entry product and
    exit step meet in ONE kernel), and three passes per transform (2^21); the two-pass sizes are covered above."""
    a, b, c = orc.rand_fr(11, n), orc.rand_fr(12, n), orc.rand_fr(13, n)
    assert (zk.compute_h(a, b, c, log_n) == orc.groth16_compute_h(a, b, c, log_n)).all()


def test_groth16_golden_proofs(golden):
    """Byte-identical 128-B proofs on the committed instances (each verified by the independent pairing check when
    the fixture was generated), incl. the reference's own toy circuit X*Y=Z (main.go:80-107)."""
    for e in golden["groth16"]:
        pkd = golden_pk(e)
        pk = zk.ProvingKey(pkd["log_domain"], pkd["n_wires"], pkd["n_public"], pkd["g1_alpha"], pkd["g1_beta"], pkd["g1_delta"], pkd["g1_a"],
                           pkd["g1_b"], pkd["g1_k"], pkd["g1_z"], pkd["g2_beta"], pkd["g2_delta"], pkd["g2_b"])
        a, b, c, w = (mont_limbs([h2i(v) for v in e[k]]) for k in ("a", "b", "c", "w"))
        r, s = mont_limbs([h2i(e["r"])])[0], mont_limbs([h2i(e["s"])])[0]
        assert zk.prove(pk, a, b, c, w, r, s).hex() == e["proof"], e["name"]
        pk.free()


@pytest.mark.parametrize("log_n,witness,tables", [(10, False, True), (10, False, False), (14, True, True), (14, True, False)])
def test_groth16_prove_vs_oracle_random_pk(log_n, witness, tables):
    """Synthetic proving key (random valid bases, like bench.py's workload): GPU proof bytes == oracle proof bytes."""
    N = 1 << log_n
    n_wires, n_public = N - 3, 5
    pkd = dict(log_domain=log_n, n_wires=n_wires, n_public=n_public,
               g1_alpha=orc.g1_gen_points(1, 1)[0], g1_beta=orc.g1_gen_points(2, 1)[0], g1_delta=orc.g1_gen_points(3, 1)[0],
               g1_a=orc.g1_gen_points(4, n_wires), g1_b=orc.g1_gen_points(5, n_wires), g1_k=orc.g1_gen_points(6, n_wires - n_public),
               g1_z=orc.g1_gen_points(7, N), g2_beta=orc.g2_gen_points(8, 1)[0], g2_delta=orc.g2_gen_points(9, 1)[0],
               g2_b=orc.g2_gen_points(10, n_wires))
    pkd["g1_a"][7] = 0  # points at infinity in A / B (gnark's InfinityA / InfinityB case)
    pkd["g1_b"][11] = 0
    pkd["g2_b"][11] = 0
    n_cons = N - 10
    a, b = orc.rand_fr(20, n_cons), orc.rand_fr(21, n_cons)
    c = np.stack([orc.fe_op("mul", 0, a[i], b[i]) for i in range(n_cons)])
    w = orc.rand_fr(22, n_wires, witness_like=witness)
    r, s = orc.rand_fr(23, 1)[0], orc.rand_fr(24, 1)[0]
    exp, _ = orc.groth16_prove(pkd, a, b, c, w, r, s)
    pk = zk.ProvingKey(**pkd, precompute_tables=tables)  # resident-key window tables on / off: same bytes either way
    assert zk.prove(pk, a, b, c, w, r, s) == exp
    assert zk.prove(pk, a, b, c, w, r, s) == exp  # the key is reusable
    pk.free()


def test_sharded_prove_path_matches_single_call():
    """The multi-GPU decomposition (zk_bn254_groth16_msm5_dev on slices + all-gather + zk_bn254_groth16_finalize), executed here
    as TWO slices on one GPU, yields the same 128 bytes as the single-call prover and the oracle."""
    from noir_backend_using_gnark_amd import parallel as par
    log_n = 11
    N = 1 << log_n
    nw, npub = N, 4
    pkd = dict(log_domain=log_n, n_wires=nw, n_public=npub,
               g1_alpha=orc.g1_gen_points(1, 1)[0], g1_beta=orc.g1_gen_points(2, 1)[0], g1_delta=orc.g1_gen_points(3, 1)[0],
               g1_a=orc.g1_gen_points(4, nw), g1_b=orc.g1_gen_points(5, nw), g1_k=orc.g1_gen_points(6, nw - npub),
               g1_z=orc.g1_gen_points(7, N), g2_beta=orc.g2_gen_points(8, 1)[0], g2_delta=orc.g2_gen_points(9, 1)[0],
               g2_b=orc.g2_gen_points(10, nw))
    a, b = orc.rand_fr(20, N), orc.rand_fr(21, N)
    c = np.stack([orc.fe_op("mul", 0, a[i], b[i]) for i in range(N)])
    w = orc.rand_fr(22, nw, witness_like=True)
    r, s = orc.rand_fr(23, 1)[0], orc.rand_fr(24, 1)[0]
    exp, _ = orc.groth16_prove(pkd, a, b, c, w, r, s)
    pk = zk.ProvingKey(**pkd)
    assert zk.prove(pk, a, b, c, w, r, s) == exp
    # device-resident copies of everything, K indexed by wire like bench.py does
    D = _lib.DeviceBuffer.from_numpy
    d_a, d_b, d_z, d_b2, d_w = D(pkd["g1_a"]), D(pkd["g1_b"]), D(pkd["g1_z"]), D(pkd["g2_b"]), D(w)
    k_wire = np.zeros((nw, 8), np.uint64)
    k_wire[npub:] = pkd["g1_k"]
    d_k = D(k_wire)
    d_h = _lib.DeviceBuffer(N * 32)
    da, db, dc = D(a), D(b), D(c)
    _lib.check(_lib.lib().zk_bn254_groth16_compute_h_dev(C.c_void_p(da.ptr), C.c_void_p(db.ptr), C.c_void_p(dc.ptr), C.c_size_t(N),
                                                         C.c_uint32(log_n), C.c_void_p(d_h.ptr), None))
    recs = []
    for rank, world in ((0, 2), (1, 2)):
        lo, hi = par.shard_range(nw, rank, world)
        k_skip = npub if rank == 0 else 0
        nz = (hi - lo) - (1 if rank == world - 1 else 0)
        recs.append(par.groth16_msm5_local(d_a.ptr + lo * 64, d_b.ptr + lo * 64, d_b2.ptr + lo * 128, d_w.ptr + lo * 32, hi - lo,
                                           d_k.ptr + (lo + k_skip) * 64, d_w.ptr + (lo + k_skip) * 32, hi - lo - k_skip,
                                           d_z.ptr + lo * 64, d_h.ptr + lo * 32, nz))
    assert par.groth16_finalize(pk, np.stack(recs), r, s) == exp
    pk.free()


def test_device_pointer_transforms_and_compute_h_dev():
    """*_dev entry points (data resident in HBM): NTT, BitReverse, computeH against the oracle."""
    log_n = 13
    n = 1 << log_n
    x = orc.rand_fr(77, n)
    d = _lib.DeviceBuffer.from_numpy(x)
    L = _lib.lib()
    _lib.check(L.zk_bn254_ntt_dev(C.c_void_p(d.ptr), C.c_uint32(log_n), C.c_int(1), C.c_int(zk.DIF), C.c_int(1), None))
    assert (d.to_numpy(np.uint64, (n, 4)) == orc.fr_ntt(x, True, orc.DIF, True)).all()
    _lib.check(L.zk_bn254_bit_reverse_dev(C.c_void_p(d.ptr), C.c_uint32(log_n), None))
    assert (d.to_numpy(np.uint64, (n, 4)) == orc.fr_bit_reverse(orc.fr_ntt(x, True, orc.DIF, True))).all()
    a, b, c = orc.rand_fr(1, n - 5), orc.rand_fr(2, n - 5), orc.rand_fr(3, n - 5)
    da, db, dc, dh = (_lib.DeviceBuffer.from_numpy(v) for v in (a, b, c)), None, None, None
    da, db, dc = da
    dh = _lib.DeviceBuffer(n * 32)
    _lib.check(L.zk_bn254_groth16_compute_h_dev(C.c_void_p(da.ptr), C.c_void_p(db.ptr), C.c_void_p(dc.ptr), C.c_size_t(n - 5), C.c_uint32(log_n),
                                                C.c_void_p(dh.ptr), None))
    assert (dh.to_numpy(np.uint64, (n, 4)) == orc.groth16_compute_h(a, b, c, log_n)).all()
    assert (da.to_numpy(np.uint64, (n - 5, 4)) == a).all()  # inputs untouched
    with pytest.raises(_lib.ZkmiError):
        _lib.check(L.zk_bn254_ntt_dev(C.c_void_p(d.ptr), C.c_uint32(29), C.c_int(0), C.c_int(zk.DIF), C.c_int(0), None))


@pytest.mark.parametrize("c", [0, 8, 16])
def test_msm_equal_and_opposite_bucket_sums(c):
    """Adjacent buckets holding the SAME point (running sum doubles) and OPPOSITE points (running sum returns to infinity):
    drives the same-x branches of the XYZZ + XYZZ addition in the bucket-reduction tail (29-bit form on G1)."""
    P = orc.g1_gen_points(71, 1)[0]
    negP = P.copy()
    y = from_mont_limbs(P[4:8], ref.Q)[0]
    negP[4:8] = mont_limbs([(ref.Q - y) % ref.Q], ref.Q)[0]
    rest_p, rest_s = orc.g1_gen_points(72, 300), ref.rand_felts(73, 300)
    for d in (5, 77, 200):
        for second, pt2 in ((d + 1, P), (d + 1, negP), (d, negP)):
            pts = np.concatenate([np.stack([P, pt2]), rest_p])
            sc = mont_limbs([d, second] + rest_s)
            cfg = zk.MultiExpConfig(scalars_mont=True, window_bits=c)
            assert (zk.g1_multi_exp(pts, sc, cfg) == orc.g1_msm(pts, sc)).all(), (c, d, second)
    # G2: same construction
    Pg = orc.g2_gen_points(74, 1)[0]
    negPg = Pg.copy()
    ys = from_mont_limbs(Pg[8:16], ref.Q)
    negPg[8:16] = mont_limbs([(ref.Q - v) % ref.Q for v in ys], ref.Q).reshape(-1)
    rp, rs = orc.g2_gen_points(75, 100), ref.rand_felts(76, 100)
    for second, pt2 in ((8, Pg), (8, negPg), (7, negPg)):
        pts = np.concatenate([np.stack([Pg, pt2]), rp])
        sc = mont_limbs([7, second] + rs)
        assert (zk.g2_multi_exp(pts, sc, zk.MultiExpConfig(scalars_mont=True, window_bits=c)) == orc.g2_msm(pts, sc)).all(), (c, second)


def _h_phase_gpu(phase, a, b, c, log_d, log_g, rank):
    """One compute phase of the sharded computeH through the C ABI, in place on numpy blocks (M, 4) of Montgomery limbs."""
    D = _lib.DeviceBuffer.from_numpy
    da, db, dc = D(a), (D(b) if b is not None else None), (D(c) if c is not None else None)
    _lib.check(_lib.lib().zk_bn254_groth16_h_shard_dev(C.c_int(phase), C.c_void_p(da.ptr), C.c_void_p(db.ptr if db else 0),
                                                       C.c_void_p(dc.ptr if dc else 0), C.c_uint32(log_d), C.c_uint32(log_g),
                                                       C.c_uint32(rank), None))
    for dst, src in ((a, da), (b, db), (c, dc)):
        if dst is not None:
            dst[...] = src.to_numpy(np.uint64, dst.shape)


@pytest.mark.parametrize("log_d,G", [(6, 8), (9, 2), (12, 1), (13, 4), (15, 8), (18, 8), (20, 2)])
def test_sharded_compute_h_matches_single_gpu_and_oracle(log_d, G):
    """Block-sharded computeH (4 phases + transposes, played for all G ranks in lock-step on one GPU) == computeH on one GPU,
    bit for bit; small sizes also against the oracle."""
    from tests import sharded_h_ref as sh
    n = 1 << log_d
    M = n // G
    a, b, c = orc.rand_fr(51, n), orc.rand_fr(52, n), orc.rand_fr(53, n)
    want = zk.compute_h(a, b, c, log_d)
    if log_d <= 15:
        assert (want == orc.groth16_compute_h(a, b, c, log_d)).all()
    blk = lambda v: [v[r * M:(r + 1) * M].copy() for r in range(G)]
    H = sh.run_virtual(_h_phase_gpu, blk(a), blk(b), blk(c), log_d)
    got = np.concatenate(H)
    assert sha_image(got) == sha_image(want)
    if log_d in (9, 13, 15):   # the pipelined schedule's per-array calls (phases 0, 1, 4 with b = c = NULL, then 5)
        H = sh.run_virtual(_h_phase_gpu, blk(a), blk(b), blk(c), log_d, per_array=True)
        assert sha_image(np.concatenate(H)) == sha_image(want)
    # the six-transform schedule (c in coefficient form: phases 0, 1, 4, 6, 7, 8; 9 transposes) -- the default of parallel.compute_h_sharded
    H = sh.run_virtual_six(_h_phase_gpu, blk(a), blk(b), blk(c), log_d)
    assert sha_image(np.concatenate(H)) == sha_image(want)


def _ntt_step_gpu(step, a, log_d, log_g, rank, inverse, decimation, coset):
    """One step of the sharded standalone transform through the C ABI, in place on a numpy block (M, 4)."""
    da = _lib.DeviceBuffer.from_numpy(a)
    _lib.check(_lib.lib().zk_bn254_ntt_shard_dev(C.c_int(step), C.c_void_p(da.ptr), C.c_uint32(log_d), C.c_uint32(log_g), C.c_uint32(rank), C.c_int(int(inverse)),
                                                 C.c_int(int(decimation)), C.c_int(int(coset)), None))
    a[...] = da.to_numpy(np.uint64, a.shape)


@pytest.mark.parametrize("log_d,G", [(6, 8), (10, 2), (12, 1), (13, 4), (15, 8), (20, 8), (21, 4)])
def test_sharded_standalone_ntt_matches_single_gpu_all_modes(log_d, G):
    """zk_bn254_ntt_shard_dev (BASELINE configs[4] on several GPUs): the steps of parallel.ntt_sharded played for all G ranks in lock-step on one GPU give
    the single-GPU (*Domain).FFT / FFTInverse bit for bit, for all eight mode combinations (the four computeH uses at the two largest sizes); the smallest
    size also against the oracle."""
    from tests import sharded_h_ref as sh
    n = 1 << log_d
    M = n // G
    x = orc.rand_fr(600 + log_d, n)
    dom = zk.Domain(n)
    modes = [(i, d, c) for i in (0, 1) for d in (zk.DIT, zk.DIF) for c in (0, 1)]
    if log_d >= 20:
        modes = [(0, zk.DIF, 1), (1, zk.DIF, 0), (0, zk.DIT, 1), (1, zk.DIT, 1)]
    for inverse, dec, coset in modes:
        want = x.copy()
        (dom.fft_inverse if inverse else dom.fft)(want, dec, bool(coset))
        if log_d == 6:
            assert (want == orc.fr_ntt(x, bool(inverse), dec, bool(coset))).all()
        blocks = [x[r * M:(r + 1) * M].copy() for r in range(G)]
        out = sh.run_virtual_ntt(_ntt_step_gpu, blocks, log_d, bool(inverse), dec, bool(coset))
        assert sha_image(np.concatenate(out)) == sha_image(want), (log_d, G, inverse, dec, coset)
    d = _lib.DeviceBuffer(64 * 32)
    call = lambda st, ld, lg, rk, dec=zk.DIF: _lib.lib().zk_bn254_ntt_shard_dev(C.c_int(st), C.c_void_p(d.ptr), C.c_uint32(ld), C.c_uint32(lg), C.c_uint32(rk), C.c_int(0),
                                                                             C.c_int(dec), C.c_int(0), None)
    assert call(3, 6, 1, 0) == _lib.ZK_ERR_ARG and call(0, 6, 4, 0) == _lib.ZK_ERR_ARG and call(0, 6, 1, 2) == _lib.ZK_ERR_ARG and call(0, 6, 1, 0, 7) == _lib.ZK_ERR_ARG
    assert call(0, 3, 2, 0) == _lib.ZK_ERR_ARG  # fewer than G columns per rank


def test_sharded_compute_h_argument_errors():
    d = _lib.DeviceBuffer(64 * 32)
    call = lambda ph, ld, lg, rk: _lib.lib().zk_bn254_groth16_h_shard_dev(C.c_int(ph), C.c_void_p(d.ptr), C.c_void_p(d.ptr), C.c_void_p(d.ptr),
                                                                          C.c_uint32(ld), C.c_uint32(lg), C.c_uint32(rk), None)
    assert call(0, 6, 4, 0) != 0      # more than 8 ranks
    assert call(0, 6, 2, 4) != 0      # rank out of range
    assert call(9, 6, 1, 0) != 0      # unknown phase (0..8 exist)
    assert call(0, 3, 2, 0) != 0      # blocks smaller than the number of ranks
    out = np.zeros(96, np.uint64)
    assert _lib.lib().zk_bn254_groth16_msm5_pk_end(C.c_uint64(12345), C.c_void_p(d.ptr), _lib.vp(out), None) != 0  # unknown session


@pytest.mark.parametrize("G,tables", [(2, True), (4, True), (4, False), (8, True)])
def test_sharded_proof_with_rank_local_keys(G, tables):
    """The whole multi-GPU prover, all ranks played on one GPU: every rank loads ITS slice of the key (window tables over the
    slice), runs the sharded computeH and zk_bn254_groth16_msm5_pk; the gathered records finalize to the oracle's 128 bytes."""
    from noir_backend_using_gnark_amd import parallel as par
    from tests import sharded_h_ref as sh
    log_n = 11
    N = 1 << log_n
    nw, npub = N, 4
    M = N // G
    small = dict(g1_alpha=orc.g1_gen_points(1, 1)[0], g1_beta=orc.g1_gen_points(2, 1)[0], g1_delta=orc.g1_gen_points(3, 1)[0],
                 g2_beta=orc.g2_gen_points(8, 1)[0], g2_delta=orc.g2_gen_points(9, 1)[0])
    pkd = dict(log_domain=log_n, n_wires=nw, n_public=npub, g1_a=orc.g1_gen_points(4, nw), g1_b=orc.g1_gen_points(5, nw),
               g1_k=orc.g1_gen_points(6, nw - npub), g1_z=orc.g1_gen_points(7, N), g2_b=orc.g2_gen_points(10, nw), **small)
    a, b = orc.rand_fr(20, N), orc.rand_fr(21, N)
    c = np.stack([orc.fe_op("mul", 0, a[i], b[i]) for i in range(N)])
    w = orc.rand_fr(22, nw, witness_like=True)
    r, s = orc.rand_fr(23, 1)[0], orc.rand_fr(24, 1)[0]
    exp, _ = orc.groth16_prove(pkd, a, b, c, w, r, s)
    blk = lambda v: [v[q * M:(q + 1) * M].copy() for q in range(G)]
    H = sh.run_virtual(_h_phase_gpu, blk(a), blk(b), blk(c), log_n)
    recs, keys = [], []
    for rank in range(G):
        lo, hi = rank * M, (rank + 1) * M
        np_loc = npub if rank == 0 else 0
        pk = zk.ProvingKey(log_n - (G.bit_length() - 1), M, np_loc, g1_a=pkd["g1_a"][lo:hi], g1_b=pkd["g1_b"][lo:hi],
                           g1_k=pkd["g1_k"][lo + np_loc - npub:hi - npub], g1_z=pkd["g1_z"][lo:hi], g2_b=pkd["g2_b"][lo:hi],
                           precompute_tables=tables, shard_full_z=(rank != G - 1), **small)
        keys.append(pk)
        d_w, d_h = _lib.DeviceBuffer.from_numpy(w[lo:hi]), _lib.DeviceBuffer.from_numpy(H[rank])
        if rank % 2 == 0:
            recs.append(par.groth16_msm5_pk(pk, d_w.ptr, d_h.ptr))
        else:  # the two-call form: preparation of w first, the rest once h exists
            sess = par.groth16_msm5_pk_begin(pk, d_w.ptr)
            recs.append(par.groth16_msm5_pk_end(sess, d_h.ptr))
    assert par.groth16_finalize(keys[0], np.stack(recs), r, s) == exp
    assert par.groth16_finalize(keys[-1], np.stack(recs), r, s) == exp
    for pk in keys:
        pk.free()


def test_felt_vector_codec_golden_and_reference_literals(golden):
    """DeserializeFelts / encode_felts on the device: the committed wire vectors (which include the reference's own witness literals,
    gnark_backend_ffi/main.go:225-246: {0, 1, -1, -1, 1, 0}) decode to the Montgomery images and encode back to the same text."""
    from noir_backend_using_gnark_amd import wire
    w = golden["wire"]
    felts = [h2i(x) for x in w["felts"]]
    d, n = wire.deserialize_felts(w["encoded"])
    assert n == len(felts)
    assert (d.to_numpy(np.uint64, (n, 4)) == mont_limbs(felts)).all()
    assert wire.serialize_felts(d, n) == w["encoded"]
    # the reference's literal: -1 mod r  (main.go:233)
    minus_one = "30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000000"
    d, n = wire.deserialize_felts("00000002" + minus_one + "0" * 63 + "1")
    assert (d.to_numpy(np.uint64, (2, 4)) == mont_limbs([ref.R - 1, 1])).all()
    assert wire.deserialize_felts("00000000")[1] == 0


@pytest.mark.parametrize("n", [1, 255, 4097, 1 << 16])
def test_felt_vector_codec_roundtrip_vs_oracle(n):
    from noir_backend_using_gnark_amd import wire
    felts = ref.rand_felts(900 + n, n)
    text = ref.felts_wire(felts).hex()
    d, m = wire.deserialize_felts(text.upper() if n == 255 else text)   # hex.DecodeString accepts both cases
    assert m == n
    assert (d.to_numpy(np.uint64, (n, 4)) == mont_limbs(felts)).all()
    assert wire.serialize_felts(d, n) == text


def test_felt_vector_codec_errors():
    from noir_backend_using_gnark_amd import wire
    one = "0" * 63 + "1"
    r_hex = "%064x" % ref.R
    for bad in ("00000002" + one,                 # count says 2, one felt present
                "00000001" + one + "00",          # trailing bytes
                "00000001" + one[:-1] + "g",      # not hex
                "00000001" + one[:-1] + "@",      # not hex (a character whose low nibble would decode)
                "00000001" + r_hex,               # r itself: gnark-crypto rejects non-canonical encodings, it does not reduce
                "0000000z" + one):                # bad count
        with pytest.raises(Exception):
            wire.deserialize_felts(bad)
    # bytes that are not hex digits but whose low nibble would decode: control bytes 0x10-0x19 (which `| 0x20` folds onto '0'-'9'),
    # NUL, and bytes >= 0x80 -- hex.DecodeString rejects them all
    for byte in (0x10, 0x11, 0x19, 0x00, 0x80, 0xb5, 0xe1, 0x1a, 0x2f, 0x3a, 0x60, 0x67):
        for pos in (0, 17, 63):
            felt = bytearray(one.encode())
            felt[pos] = byte
            with pytest.raises(Exception):
                wire.deserialize_felts(b"00000001" + bytes(felt))
    d, n = wire.deserialize_felts("00000001" + "%064x" % (ref.R - 1))
    assert n == 1


def test_two_process_sharded_proof_equals_single_process():
    """The whole multi-process program (bench.py --gpus 2: one process per rank, block-sharded computeH with its exchanges, rank-local keys,
    all-gather, finalize) run as two processes sharing this box's GPU over gloo produces the proof bytes of the single-process run on the same
    global instance (2 x 2^12 constraints == 1 x 2^13)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    common = ["--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--micro-log-n", "14"]
    single = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--log-n", "13"] + common, capture_output=True, text=True, timeout=600)
    assert single.returncode == 0, single.stdout[-2000:] + single.stderr[-2000:]
    sha1 = json.loads(single.stdout.strip().splitlines()[-1])["proof_sha"]
    env = dict(os.environ, ZKMI_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    multi = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--log-n", "12"] + common,
                           capture_output=True, text=True, timeout=900, env=env)
    assert multi.returncode == 0, multi.stdout[-2000:] + multi.stderr[-2000:]
    out = json.loads([l for l in multi.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["proof_sha"] == sha1
    # configs[4] over the two ranks: the range-sharded MSM agrees on both ranks and with its odd-split recombination, the block-sharded FFT inverts
    ms = out["micro_2p14_sharded"]
    assert ms["ranks"] == 2 and ms["msm_same_on_every_rank"] and ms["msm_equals_odd_split_recombination"] and ms["ntt_inverse_of_forward_is_identity"]
    assert "parity_error" not in out
    assert out["n_ranks_seen"] == 2 and out["config"]["collectives"] == "gloo"
    # the window-sharded decomposition (whole key on every rank, table rows split, h all-gathered) gives the same bytes -- and this run is typed the way the
    # driver types its one-GPU command, `python3 bench.py --gpus 2 ...` with NO launcher and no backend variable: bench.py starts its two ranks itself
    # (bench_blocks/launch.py: a child torch.distributed.run; gloo because this box shows fewer devices than ranks)
    env_bare = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "ZKMI_DIST_BACKEND")}
    multi = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--log-n", "12", "--shard", "windows"] + common,
                           capture_output=True, text=True, timeout=900, env=env_bare)
    assert multi.returncode == 0, multi.stdout[-2000:] + multi.stderr[-2000:]
    lines = [l for l in multi.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines  # ONE JSON line, relayed from the child's rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["proof_sha"] == sha1 and "window-sharded" in out["config"]["parallelism"]
    assert out["n_ranks_seen"] == 2 and out["config"]["collectives"] == "gloo" and out["config"]["launch"]["ranks_share_gpus"] is True
    assert "torch.distributed.run" in out["config"]["launch"]["command"]


def test_rccl_executes_the_collectives_of_the_sharded_proof_world_of_one():
    """RCCL itself (torch.distributed backend "nccl") carries every collective of the multi-GPU program -- the pipelined async all_to_all_single
    transposes of computeH on the library's stream, all_gather_into_tensor of h (window mode) and of the partial-sum records -- in a world of ONE rank
    (ZKMI_FORCE_COLLECTIVES=1: RCCL refuses two ranks on one device, and this pool has one GPU per box).  What this pins on hardware: the dtypes and
    shapes RCCL is handed, the async work handles, and the stream ordering between RCCL's stream and libzkmi's kernels; the proof bytes must equal the
    single-call prover's.  What it cannot show: anything about xGMI traffic or scaling."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--log-n", "14", "--micro-log-n", "16"]
    single = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True, timeout=600)
    assert single.returncode == 0, single.stdout[-2000:] + single.stderr[-2000:]
    sha1 = json.loads(single.stdout.strip().splitlines()[-1])["proof_sha"]
    env = dict(os.environ, ZKMI_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1")
    env.pop("ZKMI_DIST_BACKEND", None)
    for shard in ("range", "windows"):
        run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-sharded", "--shard", shard] + common, capture_output=True, text=True,
                             timeout=900, env=env)
        assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
        out = json.loads([l for l in run.stdout.strip().splitlines() if l.startswith("{")][-1])
        assert out["proof_sha"] == sha1, shard
        assert out["config"].get("collectives") == "nccl", out["config"]
        ms = out["micro_2p16_sharded"]  # the standalone sharded MSM / FFT: their all-gather and the two all-to-alls per transform through RCCL as well
        assert ms["msm_same_on_every_rank"] and ms["msm_equals_odd_split_recombination"] and ms["ntt_inverse_of_forward_is_identity"]


def test_concurrent_callers_are_safe():
    """gnark issues its MultiExp calls from several goroutines at once: four host threads calling MSMs / NTTs / a proof concurrently
    (ctypes releases the GIL) must each get the single-threaded results."""
    import threading
    n = 3000
    pts, sc = orc.g1_gen_points(81, n), orc.rand_fr(82, n)
    want_msm = orc.g1_msm(pts, sc)
    x = orc.rand_fr(83, 1 << 12)
    want_ntt = orc.fr_ntt(x, False, ref.DIF)
    pts2, sc2 = orc.g2_gen_points(84, 500), orc.rand_fr(85, 500)
    want_g2 = orc.g2_msm(pts2, sc2)
    errors = []

    def worker(kind):
        try:
            for _ in range(6):
                if kind == 0:
                    assert (zk.g1_multi_exp(pts, sc, config=MONT) == want_msm).all()
                elif kind == 1:
                    assert (zk.Domain(1 << 12).fft(x.copy(), zk.DIF) == want_ntt).all()
                elif kind == 2:
                    assert (zk.g2_multi_exp(pts2, sc2, config=MONT) == want_g2).all()
                else:
                    assert (zk.g1_multi_exp(pts, sc, zk.MultiExpConfig(scalars_mont=True, window_bits=9)) == want_msm).all()
        except Exception as e:  # noqa: BLE001
            errors.append((kind, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in (0, 1, 2, 3, 0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_cpp_host_mirror_runs_msm_and_fft(tmp_path):
    """The C++ mirror of the gnark-crypto interface (include/zkmi.hpp: G1Affine::MultiExp, fft::Domain::FFT / FFTInverse) against the oracle."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "mirror_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "mirror_check.cpp"),
                           "-L" + os.path.join(root, "noir_backend_using_gnark_amd"), "-lzkmi", "-Wl,-rpath," + os.path.join(root, "noir_backend_using_gnark_amd"),
                           "-o", exe])
    n, log_n = 777, 10
    pts, sc = orc.g1_gen_points(91, n), orc.rand_fr(92, n)
    x = orc.rand_fr(93, 1 << log_n)
    blob = tmp_path / "blob.bin"
    with open(blob, "wb") as f:
        f.write(np.uint64(n).tobytes() + pts.tobytes() + sc.tobytes() + orc.g1_msm(pts, sc).tobytes())
        f.write(orc.ints_to_limbs(from_mont_limbs(sc)).tobytes())  # the same scalars in regular form (upstream's default config)
        f.write(np.uint64(log_n).tobytes() + x.tobytes() + orc.fr_ntt(x, False, ref.DIF).tobytes())
    out = subprocess.run([exe, str(blob)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr


def test_registered_bases_use_window_tables_g1_g2_offsets_and_skew():
    """zk_bn254_bases_register builds window tables for >= 4096 bases (KZG: many commits against one SRS); commits through them -- full,
    sub-range with offset, witness-like scalars, G2, explicit window width (plain method) -- all equal the oracle."""
    n = 6000
    pts = orc.g1_gen_points(101, n)
    rb = zb.ResidentBases(pts)
    for seed, wl in ((102, False), (103, True)):
        sc = orc.rand_fr(seed, n, witness_like=wl)
        assert (rb.multi_exp(sc, config=MONT) == orc.g1_msm(pts, sc)).all()
        assert (rb.multi_exp(sc[:4500], offset=1500, config=MONT) == orc.g1_msm(pts[1500:], sc[:4500])).all()
        assert (rb.multi_exp(sc[:10], offset=5990, config=MONT) == orc.g1_msm(pts[5990:], sc[:10])).all()
        assert (rb.multi_exp(sc, zk.MultiExpConfig(scalars_mont=True, window_bits=11)) == orc.g1_msm(pts, sc)).all()
    zero = np.zeros((n, 4), np.uint64)
    assert (rb.multi_exp(zero, config=MONT) == 0).all()
    rb.free()
    # bases and scalars already in HBM (kzg.Commit of a polynomial the NTTs left on the device)
    dp, ds = _lib.DeviceBuffer.from_numpy(pts), _lib.DeviceBuffer.from_numpy(sc)
    rbd = zb.ResidentBases(dp, n=n)
    assert (rbd.multi_exp_dev(ds, n, config=MONT) == orc.g1_msm(pts, sc)).all()
    assert (rbd.multi_exp_dev(ds.ptr + 100 * 32, 3000, offset=100, config=MONT) == orc.g1_msm(pts[100:3100], sc[100:3100])).all()
    rbd.free()
    p2 = orc.g2_gen_points(104, 4200)
    rb2 = zb.ResidentBases(p2, is_g2=True)
    s2 = orc.rand_fr(105, 4200)
    assert (rb2.multi_exp(s2, config=MONT) == orc.g2_msm(p2, s2)).all()
    assert (rb2.multi_exp(s2[:1000], offset=3000, config=MONT) == orc.g2_msm(p2[3000:4000], s2[:1000])).all()
    rb2.free()


# ------------------------------------------------------------------------------------------------ full-size code paths (round 2)
# The planner picks window widths 21 / 22 (window tables at >= 2^22 points, the 2^26 MSM) and raises the task size to n >> 16 above 2^21
# points; nothing below ran those paths in round 1.  Small n with the width FORCED reaches the same kernels and bucket geometry; the
# full sizes are checked against the oracle where it finishes in seconds (2^22) and through size-independent properties above that.
@pytest.mark.parametrize("c", [21, 22])
def test_msm_plain_window_bits_21_22_vs_oracle(c):
    n = 3000
    pts, sc = orc.g1_gen_points(201, n), orc.rand_fr(202, n, witness_like=(c == 22))
    assert (zk.g1_multi_exp(pts, sc, zk.MultiExpConfig(scalars_mont=True, window_bits=c)) == orc.g1_msm(pts, sc)).all()
    p2, s2 = orc.g2_gen_points(203, 400), orc.rand_fr(204, 400)
    assert (zk.g2_multi_exp(p2, s2, zk.MultiExpConfig(scalars_mont=True, window_bits=c)) == orc.g2_msm(p2, s2)).all()
    with pytest.raises(_lib.ZkmiError):
        zk.g1_multi_exp(pts, sc, zk.MultiExpConfig(scalars_mont=True, window_bits=23))


@pytest.mark.parametrize("c", [8, 21, 22, 23, 24])
def test_registered_bases_table_window_bits_vs_oracle(c):
    """window tables 2^(c*w) * P_i with the widths the planner uses at 2^22 .. 2^26 points, on few points (table_window_bits knob)."""
    n = 2500
    pts = orc.g1_gen_points(211, n)
    rb = zb.ResidentBases(pts, table_window_bits=c)
    for seed, wl in ((212, False), (213, True)):
        sc = orc.rand_fr(seed, n, witness_like=wl)
        assert (rb.multi_exp(sc, config=MONT) == orc.g1_msm(pts, sc)).all()
        assert (rb.multi_exp(sc[:700], config=MONT, offset=1800) == orc.g1_msm(pts[1800:], sc[:700])).all()
    # regular-form scalars (upstream's default config) against the same tables
    sc = orc.rand_fr(214, n)
    assert (rb.multi_exp(orc.ints_to_limbs(from_mont_limbs(sc))) == orc.g1_msm(pts, sc)).all()
    rb.free()
    p2, s2 = orc.g2_gen_points(215, 300), orc.rand_fr(216, 300)
    rb2 = zb.ResidentBases(p2, is_g2=True, table_window_bits=c)
    assert (rb2.multi_exp(s2, config=MONT) == orc.g2_msm(p2, s2)).all()
    rb2.free()
    with pytest.raises(_lib.ZkmiError):
        zb.ResidentBases(pts, table_window_bits=25)


def _random_pk(log_n, n_wires, n_public, seed0=0):
    N = 1 << log_n
    return dict(log_domain=log_n, n_wires=n_wires, n_public=n_public,
                g1_alpha=orc.g1_gen_points(seed0 + 1, 1)[0], g1_beta=orc.g1_gen_points(seed0 + 2, 1)[0], g1_delta=orc.g1_gen_points(seed0 + 3, 1)[0],
                g1_a=orc.g1_gen_points(seed0 + 4, n_wires), g1_b=orc.g1_gen_points(seed0 + 5, n_wires), g1_k=orc.g1_gen_points(seed0 + 6, n_wires - n_public),
                g1_z=orc.g1_gen_points(seed0 + 7, N), g2_beta=orc.g2_gen_points(seed0 + 8, 1)[0], g2_delta=orc.g2_gen_points(seed0 + 9, 1)[0],
                g2_b=orc.g2_gen_points(seed0 + 10, n_wires))


@pytest.mark.parametrize("c", [21, 22, 24])
def test_groth16_prove_with_table_window_bits_21_22(c):
    log_n = 10
    N = 1 << log_n
    pkd = _random_pk(log_n, N - 2, 3)
    a, b = orc.rand_fr(20, N), orc.rand_fr(21, N)
    cc = np.stack([orc.fe_op("mul", 0, a[i], b[i]) for i in range(N)])
    w = orc.rand_fr(22, N - 2, witness_like=True)
    r, s = orc.rand_fr(23, 1)[0], orc.rand_fr(24, 1)[0]
    exp, _ = orc.groth16_prove(pkd, a, b, cc, w, r, s)
    pk = zk.ProvingKey(**pkd, table_window_bits=c)
    assert pk.info() == dict(n_wires=N - 2, n_public=3, log_domain=log_n, tables=True)
    assert zk.prove(pk, a, b, cc, w, r, s) == exp
    pk.free()


def test_groth16_gnark_compact_key_layout_with_infinity_bitmaps():
    """gnark stores pk.G1.A / pk.G1.B / pk.G2.B WITHOUT their points at infinity plus the InfinityA / InfinityB bitmaps (setup.go; the
    reference would pass such a key through backend/groth16/r1cs.go:107-143): loading the compact arrays gives the same proof bytes as
    the wire-indexed arrays with (0,0) placeholders -- host arrays and device-resident arrays, tables on and off."""
    log_n = 11
    N = 1 << log_n
    nw, npub = N - 1, 4
    pkd = _random_pk(log_n, nw, npub, seed0=40)
    rng = np.random.default_rng(7)
    inf_a = rng.random(nw) < 0.3
    inf_b = rng.random(nw) < 0.45
    inf_a[:3] = [True, False, True]
    inf_b[-2:] = True
    pkd["g1_a"][inf_a] = 0
    pkd["g1_b"][inf_b] = 0
    pkd["g2_b"][inf_b] = 0
    a, b = orc.rand_fr(60, N - 7), orc.rand_fr(61, N - 7)
    cc = np.stack([orc.fe_op("mul", 0, a[i], b[i]) for i in range(N - 7)])
    w = orc.rand_fr(62, nw, witness_like=True)
    r, s = orc.rand_fr(63, 1)[0], orc.rand_fr(64, 1)[0]
    exp, _ = orc.groth16_prove(pkd, a, b, cc, w, r, s)
    compact = dict(pkd, g1_a=pkd["g1_a"][~inf_a], g1_b=pkd["g1_b"][~inf_b], g2_b=pkd["g2_b"][~inf_b])
    for tables in (True, False):
        pk = zk.ProvingKey(**compact, infinity_a=inf_a, infinity_b=inf_b, precompute_tables=tables)
        assert zk.prove(pk, a, b, cc, w, r, s) == exp
        pk.free()
    # the compact arrays already in HBM
    D = _lib.DeviceBuffer.from_numpy
    dev = dict(compact, g1_a=D(compact["g1_a"]), g1_b=D(compact["g1_b"]), g1_k=D(compact["g1_k"]), g1_z=D(compact["g1_z"]), g2_b=D(compact["g2_b"]))
    pk = zk.ProvingKey(**dev, infinity_a=inf_a, infinity_b=inf_b, bases_on_device=True)
    assert zk.prove(pk, a, b, cc, w, r, s) == exp
    pk.free()
    # a bitmap that disagrees with NbInfinity, a wrong len(w), a short array
    L = _lib.lib()
    with pytest.raises(ValueError):
        zk.ProvingKey(**compact, infinity_a=inf_a[:-1], infinity_b=inf_b)
    pk = zk.ProvingKey(**compact, infinity_a=inf_a, infinity_b=inf_b)
    with pytest.raises(ValueError, match=r"len\(w\)"):
        zk.prove(pk, a, b, cc, w[:-1], r, s)
    pk.free()
    ia8, ib8 = np.ascontiguousarray(inf_a.astype(np.uint8)), np.ascontiguousarray(inf_b.astype(np.uint8))
    ptr = lambda arr: arr.ctypes.data
    raw = _lib.Groth16PK(log_n, nw, npub, ptr(pkd["g1_alpha"]), ptr(pkd["g1_beta"]), ptr(pkd["g1_delta"]), ptr(compact["g1_a"]), ptr(compact["g1_b"]),
                         ptr(compact["g1_k"]), ptr(compact["g1_z"]), ptr(pkd["g2_beta"]), ptr(pkd["g2_delta"]), ptr(compact["g2_b"]), 0, 0,
                         ptr(ia8), ptr(ib8), int(inf_a.sum()) + 1, int(inf_b.sum()), 0, 0)
    h = C.c_uint64(0)
    assert L.zk_bn254_groth16_pk_load(C.byref(raw), C.byref(h)) == _lib.ZK_ERR_ARG  # NbInfinityA does not match the bitmap


def test_msm5_session_abort_releases_slots_and_key():
    from noir_backend_using_gnark_amd import parallel as par
    log_n = 9
    N = 1 << log_n
    pk = zk.ProvingKey(**_random_pk(log_n, N, 2))
    d_w = _lib.DeviceBuffer.from_numpy(orc.rand_fr(1, N))
    sess = par.groth16_msm5_pk_begin(pk, d_w.ptr)
    assert _lib.lib().zk_bn254_groth16_pk_free(pk.handle) == _lib.ZK_ERR_HANDLE   # a live session pins the key
    par.groth16_msm5_pk_abort(sess)
    with pytest.raises(_lib.ZkmiError):
        par.groth16_msm5_pk_abort(sess)
    # all eight slots are free again: two five-slot calls in a row must not wait
    for _ in range(2):
        s2 = par.groth16_msm5_pk_begin(pk, d_w.ptr)
        par.groth16_msm5_pk_abort(s2)
    pk.free()


def test_g1_msm_2p22_vs_oracle():
    """n = 2^22 (4x the round-1 full-size test): plain planner choice against the multi-threaded oracle; then the same points as
    registered bases with window tables (c = 21 at this size) and the two table widths forced."""
    n = 1 << 22
    L = _lib.lib()
    dp, ds = _lib.DeviceBuffer(n * 64), _lib.DeviceBuffer(n * 32)
    _lib.check(L.zk_bn254_g1_generate_dev(C.c_void_p(dp.ptr), C.c_size_t(n), C.c_uint64(0xE1), None))
    _lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(ds.ptr), C.c_size_t(n), C.c_uint64(0xE2), C.c_int(1), C.c_int(0), None))
    pts, sc = dp.to_numpy(np.uint64, (n, 8)), ds.to_numpy(np.uint64, (n, 4))
    want = orc.g1_msm(pts, sc)
    assert (zb.g1_multi_exp_dev(dp.ptr, ds.ptr, n, config=MONT) == want).all()
    for c in (0, 22):
        rb = zb.ResidentBases(dp, n=n, table_window_bits=c)
        assert (rb.multi_exp_dev(ds, n, config=MONT) == want).all(), c
        rb.free()


def _dev_scaled(d_src, n, k_limbs):
    """k * src elementwise on the device (src: n Montgomery scalars in HBM)."""
    L = _lib.lib()
    dk = _lib.DeviceBuffer.from_numpy(np.tile(k_limbs.reshape(1, 4), (n, 1)))
    out = _lib.DeviceBuffer(n * 32)
    _lib.check(L.zk_bn254_fr_mul_dev(C.c_void_p(out.ptr), C.c_void_p(d_src.ptr if hasattr(d_src, "ptr") else d_src), C.c_void_p(dk.ptr), C.c_size_t(n), None))
    dk.free()
    return out


def test_g1_msm_2p26_properties():
    """BASELINE configs[4] size (2^26 points, window width 20+ and the n >> 16 task size): homogeneity MSM(P, k*s) == k * MSM(P, s), and the
    window-table path over the same registered bases (c = 22: 48 GB of tables) equals the plain path -- two independent code paths."""
    n = 1 << 26
    L = _lib.lib()
    dp, ds = _lib.DeviceBuffer(n * 64), _lib.DeviceBuffer(n * 32)
    _lib.check(L.zk_bn254_g1_generate_dev(C.c_void_p(dp.ptr), C.c_size_t(n), C.c_uint64(0xF1), None))
    _lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(ds.ptr), C.c_size_t(n), C.c_uint64(0xF2), C.c_int(1), C.c_int(0), None))
    k = orc.rand_fr(0xF3, 1)
    dks = _dev_scaled(ds, n, k[0])
    r1 = zb.g1_multi_exp_dev(dp.ptr, ds.ptr, n, config=MONT)
    r2 = zb.g1_multi_exp_dev(dp.ptr, dks.ptr, n, config=MONT)
    assert orc.g1_on_curve(r1) and (r1 != 0).any()
    assert (orc.g1_mul(r1, k[0]) == r2).all()
    # additivity over a split at an odd position: MSM(P[:m], s[:m]) + MSM(P[m:], s[m:]) == MSM(P, s)
    m = (n // 3) | 1
    parts = np.stack([zb.g1_multi_exp_dev(dp.ptr, ds.ptr, m, config=MONT, partial=True),
                      zb.g1_multi_exp_dev(dp.ptr + m * 64, ds.ptr + m * 32, n - m, config=MONT, partial=True)])
    assert (zb.g1_sum_partials(parts) == r1).all()
    rb = zb.ResidentBases(dp, n=n)   # planner: window tables with c = 22
    assert (rb.multi_exp_dev(ds, n, config=MONT) == r1).all()
    assert (rb.multi_exp_dev(dks, n, config=MONT) == r2).all()
    rb.free()


def test_ntt_2p26_vs_oracle_identity_and_sharded_equivalence():
    """BASELINE configs[4] size, the transform: at 2^26 points (three passes, 2 GiB in HBM) FFT(DIF) and FFTInverse(DIF, coset) equal the CPU oracle's images
    (by SHA-256), FFTInverse(DIT) . FFT(DIF) is the identity on the device, and
    the 8-rank block-sharded schedule (zk_bn254_ntt_shard_dev: cross stages on transposed data, size-2^23 block transforms, two all-to-all transposes,
    radix-8 columns), played by this one GPU for all ranks, gives the single-GPU FFT(DIF) bit for bit."""
    from tests import sharded_h_ref as sh
    log_n, G = 26, 8
    n = 1 << log_n
    d = _lib.DeviceBuffer(n * 32)
    _lib.check(_lib.lib().zk_bn254_fr_random_dev(C.c_void_p(d.ptr), C.c_size_t(n), C.c_uint64(0x26), C.c_int(1), C.c_int(0), None))
    x0 = d.to_numpy(np.uint64, (n, 4))
    dom = zk.Domain(n)
    dom.fft(d, zk.DIF)
    want = sha_image(d.to_numpy(np.uint64, (n, 4)))
    dom.fft_inverse(d, zk.DIT)
    assert (d.to_numpy(np.uint64, (n, 4)) == x0).all()
    # ... and against the ORACLE at this size, once: FFT(DIF) and the mode computeH ends with, FFTInverse(DIF, coset) (orc.fr_ntt on the host cores, 2 GiB each)
    assert sha_image(orc.fr_ntt(x0, False, orc.DIF)) == want
    dom.fft_inverse(d, zk.DIF, coset=True)
    assert sha_image(d.to_numpy(np.uint64, (n, 4))) == sha_image(orc.fr_ntt(x0, True, orc.DIF, coset=True))
    d.free()
    M = n // G
    blocks = [x0[r * M:(r + 1) * M].copy() for r in range(G)]
    del x0
    out = sh.run_virtual_ntt(_ntt_step_gpu, blocks, log_n, False, zk.DIF, False)
    assert sha_image(np.concatenate(out)) == want


def test_groth16_2p20_proof_bytes_vs_oracle():
    """BASELINE configs[1]'s size inside the suite with the wire vector real circuits have: the 128 proof bytes of the synthetic 2^20-constraint instance --
    resident key with window tables, the whole single-call schedule, WITNESS-LIKE wire values (giant buckets, adaptive task lengths, zero digits dropped before the
    sort with the pair count left on the device) -- equal the bytes the CPU oracle (oracle/bn254_oracle.c, OpenMP) computes for the same key, a, b, c, w, r, s.
    (The uniform instance of this size is compared with the oracle by every bench.py run after its timed region, and at 2^24 by the test below.)"""
    import bench
    L = _lib.lib()
    inst = bench.Instance(L, _lib, zk, 20, 0, bench.N_PUBLIC, 1, True)
    got = zk.prove(inst.pk, inst.d_a, inst.d_b, inst.d_c, inst.d_w, inst.r, inst.s, n_constraints=inst.N, on_device=True)
    again = zk.prove(inst.pk, inst.d_a, inst.d_b, inst.d_c, inst.d_w, inst.r, inst.s, n_constraints=inst.N, on_device=True)
    want, _, _ = bench.oracle_proof(inst, 20)
    inst.free()
    assert got == want and again == want


def test_lagrange_form_of_a_base_array_vs_oracle():
    """zk_bn254_bases_lagrange (csrc/lagrange.hip: the inverse transform of a base array taken in the exponent + the two points of gnark's blinding): a commitment
    from EVALUATIONS against the Lagrange form equals the oracle's multi-exp of the blinded COEFFICIENTS (the oracle's FFTInverse) against the points themselves --
    for any points (the map is linear), domains of 2 .. 2^13 points, witness-like and uniform evaluations, through the plain and the window-table multi-exp, and as
    one batched call; errors as documented."""
    b = orc.rand_fr(0xD7, 2)
    for log_n in (1, 2, 5, 9, 13):
        n = 1 << log_n
        pts = orc.g1_gen_points(0xD0 + log_n, n + 3)
        rb = zb.ResidentBases(pts, table_window_bits=-1)
        lag = rb.lagrange(log_n)
        assert lag.n == n + 2
        vecs, wants = [], []
        for seed, wl in ((0xE0 + log_n, True), (0xF0 + log_n, False)):
            e = orc.rand_fr(seed, n, witness_like=wl)
            c = orc.fr_bit_reverse(orc.fr_ntt(e, True, 1))          # FFTInverse(DIF) + BitReverse: canonical coefficients
            blinded = np.concatenate([c, np.zeros((2, 4), np.uint64)])
            blinded[0] = orc.fe_op("sub", 0, blinded[0], b[0])         # + (b0 + b1 X)(X^n - 1)
            blinded[1] = orc.fe_op("sub", 0, blinded[1], b[1])
            blinded[n] = orc.fe_op("add", 0, blinded[n], b[0])
            blinded[n + 1] = orc.fe_op("add", 0, blinded[n + 1], b[1])
            want = orc.g1_msm(pts[:n + 2], blinded)
            ext = np.concatenate([e, b])
            assert (lag.multi_exp(ext, MONT) == want).all(), (log_n, wl)
            vecs.append(ext); wants.append(want)
        lag.build_table(8)
        got = lag.multi_exp_batch(vecs, config=MONT)
        assert (got[0] == wants[0]).all() and (got[1] == wants[1]).all(), log_n
        lag.free()
        with pytest.raises(_lib.ZkmiError):
            rb.lagrange(log_n + 1)          # needs 2^(log_n + 1) + 2 points
        rb.free()
    g2 = zb.ResidentBases(orc.g2_gen_points(0xD9, 6), is_g2=True)
    with pytest.raises(_lib.ZkmiError):
        g2.lagrange(2)
    g2.free()


def test_window_tables_built_after_registration():
    """zk_bn254_bases_build_table: bases registered without window tables (what the export shim does for a process's first proof) get them later; commitments
    before and after equal the oracle's, G1 and G2, planner's width and a forced one; the batched path (which needs the table) is reached afterwards; a second
    call, a small array (below 4096: nothing to build) and a bad width behave as documented."""
    n = 5000
    pts, sc = orc.g1_gen_points(0xC1, n), orc.rand_fr(0xC2, n, witness_like=True)
    want = orc.g1_msm(pts, sc)
    for c in (0, 9):
        rb = zb.ResidentBases(pts, table_window_bits=-1)
        assert (rb.multi_exp(sc, MONT) == want).all()
        rb.build_table(c)
        assert (rb.multi_exp(sc, MONT) == want).all()
        assert (rb.multi_exp(sc[:1234], MONT, offset=3000) == orc.g1_msm(pts[3000:4234], sc[:1234])).all()
        got = rb.multi_exp_batch([sc, sc[::-1].copy()], config=MONT)
        assert (got[0] == want).all() and (got[1] == orc.g1_msm(pts, sc[::-1].copy())).all()
        rb.build_table(c)
        rb.build_table(0)
        assert (rb.multi_exp(sc, MONT) == want).all()
        rb.free()
    p2, s2 = orc.g2_gen_points(0xC3, 4100), orc.rand_fr(0xC4, 4100)
    rb2 = zb.ResidentBases(p2, is_g2=True, table_window_bits=-1)
    rb2.build_table()
    assert (rb2.multi_exp(s2, MONT) == orc.g2_msm(p2, s2)).all()
    rb2.free()
    small = zb.ResidentBases(pts[:100], table_window_bits=-1)
    small.build_table()  # below 4096 bases the planner builds none
    assert (small.multi_exp(sc[:100], MONT) == orc.g1_msm(pts[:100], sc[:100])).all()
    with pytest.raises(_lib.ZkmiError):
        small.build_table(25)
    small.free()
    with pytest.raises(_lib.ZkmiError):
        small.build_table()


def test_batched_multi_exp_of_several_scalar_vectors_against_one_base_array():
    """zk_bn254_msm_bases_batch[_dev]: plonk.Prove's three simultaneous kzg.Commit calls (l, r, o; h1, h2, h3 -- gnark v0.8.0 plonk prove.go via
    backend/plonk/plonk.go:53-73) as ONE multi-scalar multiplication with a bucket set per vector.  Against the oracle and against one zk_bn254_msm_bases
    call per vector: window tables at two widths (one / three bucket-key sort passes), 2 and 3 vectors, uniform / witness-like / all-zero / equal vectors,
    offsets, device-resident scalars, regular-form scalars; the sequential fallbacks (no table, G2, explicit window width, four vectors) and the errors."""
    n = 6000
    pts = orc.g1_gen_points(0xB1, n)
    v = [orc.rand_fr(0xB2, n), orc.rand_fr(0xB3, n, witness_like=True), orc.rand_fr(0xB4, n), np.zeros((n, 4), np.uint64)]
    want = [orc.g1_msm(pts, x) for x in v[:3]]
    for tb in (0, 8, 16, -1):
        rb = zb.ResidentBases(pts, table_window_bits=tb)
        got = rb.multi_exp_batch(v[:3], config=MONT)
        assert got.shape == (3, 8)
        for k in range(3):
            assert (got[k] == want[k]).all(), (tb, k)
        got = rb.multi_exp_batch([v[1], v[0]], config=MONT)
        assert (got[0] == want[1]).all() and (got[1] == want[0]).all(), tb
        got = rb.multi_exp_batch([v[2], v[3], v[2]], config=MONT)  # an all-zero vector (an empty bucket set) between two equal ones
        assert (got[0] == want[2]).all() and (got[1] == 0).all() and (got[2] == want[2]).all(), tb
        got = rb.multi_exp_batch([x[:4500] for x in v[:3]], config=MONT, offset=1500)
        for k in range(3):
            assert (got[k] == orc.g1_msm(pts[1500:], v[k][:4500])).all(), (tb, k)
        got = rb.multi_exp_batch([x[:3] for x in v[:3]], config=MONT, offset=n - 3)  # fewer scalars than lanes of one workgroup
        for k in range(3):
            assert (got[k] == rb.multi_exp(v[k][:3], MONT, offset=n - 3)).all(), (tb, k)
        got = rb.multi_exp_batch(v[:4], config=MONT)  # four vectors: one after the other
        assert (got[3] == 0).all() and all((got[k] == want[k]).all() for k in range(3))
        got = rb.multi_exp_batch(v[:3], config=zk.MultiExpConfig(scalars_mont=True, window_bits=11))  # plain method asked for
        assert all((got[k] == want[k]).all() for k in range(3))
        assert (rb.multi_exp_batch(v[:1], config=MONT)[0] == want[0]).all()
        assert rb.multi_exp_batch([], config=MONT).shape == (0, 8)
        reg = [orc.ints_to_limbs(from_mont_limbs(x)) for x in v[:3]]  # upstream's default config: regular form
        got = rb.multi_exp_batch(reg)
        assert all((got[k] == want[k]).all() for k in range(3)), tb
        dv = [_lib.DeviceBuffer.from_numpy(x) for x in v[:3]]
        got = rb.multi_exp_batch(dv, n=n, config=MONT)
        assert all((got[k] == want[k]).all() for k in range(3)), tb
        got = rb.multi_exp_batch([d.ptr + 100 * 32 for d in dv], n=3000, config=MONT, offset=100)
        for k in range(3):
            assert (got[k] == orc.g1_msm(pts[100:3100], v[k][100:3100])).all(), (tb, k)
        with pytest.raises(ValueError):
            rb.multi_exp_batch(v[:3], config=MONT, offset=1)  # len(points) != len(scalars)
        with pytest.raises(ValueError):
            rb.multi_exp_batch([v[0], v[1][:10]], config=MONT)
        rb.free()
    p2 = orc.g2_gen_points(0xB5, 4200)
    rb2 = zb.ResidentBases(p2, is_g2=True)
    s2 = [orc.rand_fr(0xB6, 4200), orc.rand_fr(0xB7, 4200)]
    got = rb2.multi_exp_batch(s2, config=MONT)
    assert got.shape == (2, 16) and all((got[k] == orc.g2_msm(p2, s2[k])).all() for k in range(2))
    rb2.free()
    with pytest.raises(_lib.ZkmiError):
        rb2.multi_exp_batch(s2, config=MONT)  # freed handle


def test_prepared_scalars_shared_by_the_multi_exps_of_one_proof():
    """zk_bn254_scalars_register + zk_bn254_msm_bases_prepared (the inner boundary's form of groth16.Prove's A, B1, K, G2.B MultiExp calls, which pair with the
    SAME wire values: gnark v0.8.0 groth16 prove.go via main.go:131): one upload, one recoding per table geometry.  Against the oracle's sums and against
    zk_bn254_msm_bases on the same data: bases registered over the same wires share the recoding (A, B1, G2.B), K -- registered over the private wires only --
    pairs with the scalars from n_public on (a second recoding from the resident copy, no second upload), also against a K registered over ALL wires; bases
    without window tables; five concurrent threads like upstream's goroutines; length errors are upstream's."""
    import threading
    n, npub = 6000, 9
    w = orc.rand_fr(0x91, n, witness_like=True)
    pa, pb, pk_ = orc.g1_gen_points(0x92, n), orc.g1_gen_points(0x93, n), orc.g1_gen_points(0x94, n)
    p2 = orc.g2_gen_points(0x95, n)
    want = dict(a=orc.g1_msm(pa, w), b=orc.g1_msm(pb, w), k=orc.g1_msm(pk_[npub:], w[npub:]), b2=orc.g2_msm(p2, w))
    for tb in (8, -1):  # window tables (c = 8 forced: tables at this size) / none
        A, B, K = zb.ResidentBases(pa, table_window_bits=tb), zb.ResidentBases(pb, table_window_bits=tb), zb.ResidentBases(pk_[npub:], table_window_bits=tb)
        Kall, B2 = zb.ResidentBases(pk_, table_window_bits=tb), zb.ResidentBases(p2, is_g2=True, table_window_bits=tb)
        S = zb.PreparedScalars(w, MONT)
        got = dict(a=A.multi_exp_prepared(S), b=B.multi_exp_prepared(S), k=K.multi_exp_prepared(S, skip=npub), b2=B2.multi_exp_prepared(S))
        for key in want:
            assert (got[key] == want[key]).all(), (tb, key)
        assert (Kall.multi_exp_prepared(S, skip=npub, offset=npub) == want["k"]).all(), tb
        assert (A.multi_exp_prepared(S) == A.multi_exp(w, MONT)).all()
        assert (A.multi_exp_prepared(S, skip=100, offset=40) == A.multi_exp(w[100:], MONT, offset=40)).all()
        res = {}
        jobs = [("a", A, 0, 0), ("b", B, 0, 0), ("k", K, npub, 0), ("b2", B2, 0, 0), ("k2", Kall, npub, npub)]
        S2 = zb.PreparedScalars(w, MONT)  # fresh: the five arrive together at an unprepared handle
        th = [threading.Thread(target=lambda j=j: res.__setitem__(j[0], j[1].multi_exp_prepared(S2, skip=j[2], offset=j[3]))) for j in jobs]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert all((res[k2] == want[k2 if k2 != "k2" else "k"]).all() for k2 in res) and len(res) == 5, tb
        with pytest.raises(ValueError):
            K.multi_exp_prepared(S)            # n scalars against n - n_public bases: upstream's length error
        with pytest.raises(ValueError):
            A.multi_exp_prepared(S, skip=n + 1)
        for h in (S, S2, A, B, K, Kall, B2):
            h.free()
    # regular-form scalars (upstream's zero-value config) are registered in that form
    w_reg = orc.rand_fr(0x96, 300)
    A = zb.ResidentBases(pa[:300], table_window_bits=8)
    S = zb.PreparedScalars(w_reg, zk.MultiExpConfig())
    assert (A.multi_exp_prepared(S) == A.multi_exp(w_reg, zk.MultiExpConfig())).all()
    S.free()
    A.free()


def test_prepared_scalars_outlive_new_geometries_frees_and_many_registrations():
    """Lifetimes of zk_bn254_scalars_register's recodings (round 4's advisor: a third table geometry rebuilt ALL recodings in place while another caller's
    accumulate kernel, already past the lock, was still reading the old ones; a registration pinned one of the entry's eight stream slots).  Now every recoding
    is an allocation of its own held by whoever reads it: (i) readers of two geometries keep running while other threads ask for a third, a fourth and a fifth
    geometry (tables of other widths, other first scalars) on the same handle -- every result equals the oracle's; (ii) zk_bn254_scalars_free while multi-exps
    are in flight: they finish with the right sums, later calls get the unknown-handle error; (iii) twelve live registrations next to a Groth16 proof (which
    takes five slots at once)."""
    import threading
    n = 5000
    w = orc.rand_fr(0xA1, n, witness_like=True)
    pts = orc.g1_gen_points(0xA2, n)
    bases = {c: zb.ResidentBases(pts, table_window_bits=c) for c in (8, 9, 10, 11)}
    want_all = orc.g1_msm(pts, w)
    want_skip = {sk: orc.g1_msm(pts[: n - sk], w[sk:]) for sk in (7, 30)}
    S = zb.PreparedScalars(w, MONT)
    errs, done = [], []

    def reader(c, skip, reps):
        try:
            for _ in range(reps):
                got = bases[c].multi_exp_prepared(S, skip=skip)
                ok = (got == (want_all if skip == 0 else want_skip[skip])).all()
                done.append(ok)
                if not ok:
                    errs.append((c, skip))
        except Exception as e:  # noqa: BLE001
            errs.append((c, skip, repr(e)))

    th = [threading.Thread(target=reader, args=(8, 0, 12)), threading.Thread(target=reader, args=(9, 0, 12)),
          threading.Thread(target=reader, args=(10, 0, 6)), threading.Thread(target=reader, args=(11, 7, 6)), threading.Thread(target=reader, args=(8, 30, 6))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs and len(done) == 42 and all(done), errs
    # (ii) free under readers: calls that already hold the handle's data finish; the handle itself is gone at once
    S2 = zb.PreparedScalars(w, MONT)
    assert (bases[8].multi_exp_prepared(S2) == want_all).all()
    res = []
    s2_handle = C.c_uint64(S2.handle.value)

    def late_reader():
        try:
            out = np.zeros(8, dtype=np.uint64)
            rc = _lib.lib().zk_bn254_msm_bases_prepared(bases[9].handle, C.c_size_t(0), s2_handle, C.c_size_t(0), C.byref(MONT._c()), _lib.vp(out))
            res.append("gone" if rc == _lib.ZK_ERR_HANDLE else bool(rc == 0 and (out == want_all).all()))
        except Exception as e:  # noqa: BLE001
            res.append(repr(e))

    th = [threading.Thread(target=late_reader) for _ in range(4)]
    for t in th:
        t.start()
    S2.free()
    for t in th:
        t.join()
    # each of the four either got the handle's data before the free took the handle away (right sum) or finds the handle gone: never a wrong sum, never a fault
    assert len(res) == 4 and all(r is True or r == "gone" for r in res), res
    assert _lib.lib().zk_bn254_msm_bases_prepared(bases[8].handle, C.c_size_t(0), s2_handle, C.c_size_t(0), C.byref(MONT._c()), _lib.vp(np.zeros(8, np.uint64))) == _lib.ZK_ERR_HANDLE
    # (iii) more live registrations than the entry has stream slots, next to a proof
    many = [zb.PreparedScalars(w[: 100 + k], MONT) for k in range(12)]
    log_n = 8
    N = 1 << log_n
    pkd = dict(log_domain=log_n, n_wires=N, n_public=3, g1_alpha=orc.g1_gen_points(1, 1)[0], g1_beta=orc.g1_gen_points(2, 1)[0], g1_delta=orc.g1_gen_points(3, 1)[0],
               g1_a=orc.g1_gen_points(4, N), g1_b=orc.g1_gen_points(5, N), g1_k=orc.g1_gen_points(6, N - 3), g1_z=orc.g1_gen_points(7, N),
               g2_beta=orc.g2_gen_points(8, 1)[0], g2_delta=orc.g2_gen_points(9, 1)[0], g2_b=orc.g2_gen_points(10, N))
    a, b = orc.rand_fr(20, N), orc.rand_fr(21, N)
    c_ = np.stack([orc.fe_op("mul", 0, a[i], b[i]) for i in range(N)])
    ww = orc.rand_fr(22, N, witness_like=True)
    r, s_ = orc.rand_fr(23, 1)[0], orc.rand_fr(24, 1)[0]
    exp, _ = orc.groth16_prove(pkd, a, b, c_, ww, r, s_)
    pk = zk.ProvingKey(**pkd)
    assert zk.prove(pk, a, b, c_, ww, r, s_) == exp
    assert (bases[10].multi_exp_prepared(many[11], offset=0) == orc.g1_msm(pts[:111], w[:111])).all()
    pk.free()
    for h in many + [S] + list(bases.values()):
        h.free()


def test_groth16_2p24_proof_bytes_vs_oracle():
    """The metric's second size against the ORACLE, not against the library itself (BASELINE.json metric "at 2^20 / 2^24 constraints"): the 128 proof bytes of
    the synthetic 2^24-constraint instance bench.py measures -- resident key with c = 22 window tables (84 GB), the whole single-call schedule -- equal the
    bytes oracle/bn254_oracle.c computes for the same key, a, b, c, w, r, s on the host cores (100-130 s on 128 cores; ~7 GB of key and vectors come down
    from HBM for it)."""
    import bench
    L = _lib.lib()
    inst = bench.Instance(L, _lib, zk, 24, 0, bench.N_PUBLIC, 0, True)
    assert inst.pk.info()["tables"]
    got = zk.prove(inst.pk, inst.d_a, inst.d_b, inst.d_c, inst.d_w, inst.r, inst.s, n_constraints=inst.N, on_device=True)
    want, cpu_s, cores = bench.oracle_proof(inst, 24)
    inst.free()
    print("oracle: %.1f s on %d cores" % (cpu_s, cores))
    assert got == want


def test_g1_msm_2p26_vs_oracle():
    """BASELINE configs[4]'s MSM against the ORACLE: 2^26 generated points and uniform scalars come down from HBM (6 GB) and orc.g1_msm sums them on the
    host cores once; the plain path (chunked planner at this size) and the registered-bases path with its window table must both give that point."""
    n = 1 << 26
    L = _lib.lib()
    dp, ds = _lib.DeviceBuffer(n * 64), _lib.DeviceBuffer(n * 32)
    _lib.check(L.zk_bn254_g1_generate_dev(C.c_void_p(dp.ptr), C.c_size_t(n), C.c_uint64(0x26E1), None))
    _lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(ds.ptr), C.c_size_t(n), C.c_uint64(0x26E2), C.c_int(1), C.c_int(0), None))
    got = zb.g1_multi_exp_dev(dp.ptr, ds.ptr, n, config=MONT)
    rb = zb.ResidentBases(dp, n=n)
    got_table = rb.multi_exp_dev(ds, n, config=MONT)
    rb.free()
    pts, sc = dp.to_numpy(np.uint64, (n, 8)), ds.to_numpy(np.uint64, (n, 4))
    dp.free()
    ds.free()
    want = orc.g1_msm(pts, sc)
    assert (got == want).all() and (got_table == want).all()


def test_groth16_2p24_properties():
    """BASELINE configs[2] size on one GPU (2^24 constraints; key with c = 22 window tables, 84 GB): (1) the single-call prover's bytes equal
    the recombination of two half-size slices run through the table-less msm5 path (other window width, Horner, other task sizes) and
    zk_bn254_groth16_finalize; (2) homogeneity of the five sums through the table path: msm5(k*w, k*h) == k * msm5(w, h)."""
    from noir_backend_using_gnark_amd import parallel as par
    log_n = 24
    N = 1 << log_n
    npub = 8
    L = _lib.lib()

    def gen(fn, n, esz, seed):
        b = _lib.DeviceBuffer(n * esz)
        _lib.check(fn(C.c_void_p(b.ptr), C.c_size_t(n), C.c_uint64(seed), None))
        return b

    g1_a, g1_b, g1_k, g1_z = (gen(L.zk_bn254_g1_generate_dev, N, 64, sd) for sd in (0xA1, 0xB1, 0xC1, 0xD1))
    g2_b = gen(L.zk_bn254_g2_generate_dev, N, 128, 0xB2)
    small = {kk: gen(L.zk_bn254_g1_generate_dev, 1, 64, sd).to_numpy(np.uint64, (8,)) for kk, sd in (("alpha", 1), ("beta", 2), ("delta", 3))}
    small2 = {kk: gen(L.zk_bn254_g2_generate_dev, 1, 128, sd).to_numpy(np.uint64, (16,)) for kk, sd in (("beta", 8), ("delta", 9))}

    def rnd(seed, n, wit=0):
        b = _lib.DeviceBuffer(n * 32)
        _lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(b.ptr), C.c_size_t(n), C.c_uint64(seed), C.c_int(1), C.c_int(wit), None))
        return b

    d_a, d_b, d_w = rnd(0xA, N), rnd(0xB, N), rnd(0xC, N, 1)
    d_c = _lib.DeviceBuffer(N * 32)
    _lib.check(L.zk_bn254_fr_mul_dev(C.c_void_p(d_c.ptr), C.c_void_p(d_a.ptr), C.c_void_p(d_b.ptr), C.c_size_t(N), None))
    rs = orc.rand_fr(0x23, 2)
    r, s = rs[0], rs[1]
    pk = zk.ProvingKey(log_n, N, npub, small["alpha"], small["beta"], small["delta"], g1_a, g1_b, g1_k.ptr + npub * 64, g1_z,
                       small2["beta"], small2["delta"], g2_b, bases_on_device=True)
    assert pk.info()["tables"]
    proof = zk.prove(pk, d_a, d_b, d_c, d_w, r, s, n_constraints=N, on_device=True)
    # (1) two slices through the plain path
    d_h = _lib.DeviceBuffer(N * 32)
    _lib.check(L.zk_bn254_groth16_compute_h_dev(C.c_void_p(d_a.ptr), C.c_void_p(d_b.ptr), C.c_void_p(d_c.ptr), C.c_size_t(N), C.c_uint32(log_n),
                                                C.c_void_p(d_h.ptr), None))
    recs = []
    for rank in (0, 1):
        lo, hi = rank * N // 2, (rank + 1) * N // 2
        skip = npub if rank == 0 else 0
        nz = (hi - lo) - (1 if rank == 1 else 0)
        recs.append(par.groth16_msm5_local(g1_a.ptr + lo * 64, g1_b.ptr + lo * 64, g2_b.ptr + lo * 128, d_w.ptr + lo * 32, hi - lo,
                                           g1_k.ptr + (lo + skip) * 64, d_w.ptr + (lo + skip) * 32, hi - lo - skip,
                                           g1_z.ptr + lo * 64, d_h.ptr + lo * 32, nz))
    assert par.groth16_finalize(pk, np.stack(recs), r, s) == proof
    # (2) homogeneity through the key's window tables
    k = orc.rand_fr(0x77, 1)
    rec1 = par.groth16_msm5_pk(pk, d_w.ptr, d_h.ptr)
    d_kw, d_kh = _dev_scaled(d_w, N, k[0]), _dev_scaled(d_h, N, k[0])
    rec2 = par.groth16_msm5_pk(pk, d_kw.ptr, d_kh.ptr)
    for i in range(4):
        p1, p2 = zb.g1_sum_partials(rec1[16 * i:16 * i + 16]), zb.g1_sum_partials(rec2[16 * i:16 * i + 16])
        assert (p1 != 0).any() and (orc.g1_mul(p1, k[0]) == p2).all(), i
    q1, q2 = zb.g2_sum_partials(rec1[64:96]), zb.g2_sum_partials(rec2[64:96])
    assert (orc.g2_mul(q1, k[0]) == q2).all()
    # and the table path's sums are the plain path's sums
    assert (zb.g1_sum_partials(np.stack([recs[0][:16], recs[1][:16]])) == zb.g1_sum_partials(rec1[:16])).all()
    pk.free()


def _window_scalars(vals, c, rank, world):
    """gnark's signed c-bit recoding of each scalar, restricted to the windows w = rank, rank + world, ...: s^(rank) = sum_w d_w 2^(c w) mod r.
    The per-rank scalars add up to the scalar itself -- the partial MSM a window-sharded rank must return is MSM(P, s^(rank))."""
    Wd = (255 + c - 1) // c
    half = 1 << (c - 1)
    out = []
    for s in vals:
        carry, acc = 0, 0
        for w in range(Wd):
            d = ((s >> (w * c)) & ((1 << c) - 1)) + carry
            carry = 0
            if d > half:
                d -= 1 << c
                carry = 1
            if w >= rank and (w - rank) % world == 0:
                acc += d << (w * c)
        assert carry == 0
        out.append(acc % ref.R)
    return out


@pytest.mark.parametrize("G,c", [(2, 9), (4, 13), (8, 11), (8, 22)])
def test_window_sharded_keys_partial_sums_and_proof(G, c):
    """Window (table-row) sharded proving keys -- what BASELINE's north_star / configs[2] name: every rank holds the whole key but only the rows
    w = rank + k*G of its window tables.  Each rank's five partial sums equal the oracle's MSMs over the scalars restricted to that rank's
    digit windows, and the gathered records finalize to the single-GPU / oracle proof bytes."""
    from noir_backend_using_gnark_amd import parallel as par
    log_n = 9
    N = 1 << log_n
    nw, npub = N - 1, 3
    pkd = _random_pk(log_n, nw, npub, seed0=70)
    a, b = orc.rand_fr(80, N), orc.rand_fr(81, N)
    cc = np.stack([orc.fe_op("mul", 0, a[i], b[i]) for i in range(N)])
    w = orc.rand_fr(82, nw, witness_like=True)
    r, s = orc.rand_fr(83, 1)[0], orc.rand_fr(84, 1)[0]
    exp, _ = orc.groth16_prove(pkd, a, b, cc, w, r, s)
    h = zk.compute_h(a, b, cc, log_n)
    d_w, d_h = _lib.DeviceBuffer.from_numpy(w), _lib.DeviceBuffer.from_numpy(h)
    w_int, h_int = from_mont_limbs(w), from_mont_limbs(h[:N - 1])
    recs, keys = [], []
    for rank in range(G):
        assert par.window_rows(c, rank, G) == [x for x in range((255 + c - 1) // c) if x % G == rank]
        pk = zk.ProvingKey(**pkd, window_shard=(rank, G), table_window_bits=c)
        keys.append(pk)
        rec = par.groth16_msm5_pk(pk, d_w.ptr, d_h.ptr)
        recs.append(rec)
        if G <= 4 or rank in (0, G - 1):   # per-rank partial sums against the oracle (A and Z: wire scalars and h scalars)
            wr, hr = mont_limbs(_window_scalars(w_int, c, rank, G)), mont_limbs(_window_scalars(h_int, c, rank, G))
            assert (zb.g1_sum_partials(rec[0:16]) == orc.g1_msm(pkd["g1_a"], wr)).all(), rank
            assert (zb.g1_sum_partials(rec[48:64]) == orc.g1_msm(pkd["g1_z"][:N - 1], hr)).all(), rank
            assert (zb.g2_sum_partials(rec[64:96]) == orc.g2_msm(pkd["g2_b"], wr)).all(), rank
    assert par.groth16_finalize(keys[0], np.stack(recs), r, s) == exp
    assert par.groth16_finalize(keys[-1], np.stack(recs), r, s) == exp
    with pytest.raises(_lib.ZkmiError):
        zk.ProvingKey(**pkd, window_shard=(G, G))
    with pytest.raises(_lib.ZkmiError):
        zk.ProvingKey(**pkd, window_shard=(0, G), precompute_tables=False)
    for pk in keys:
        pk.free()


# ------------------------------------------------------------------------------------------------ R1CS: Setup and the solver step on the device
def _oracle_r1cs_to_device(r1):
    cons = [tuple({w: mont_limbs([cf])[0] for w, cf in lin.items()} for lin in con) for con in r1.constraints]
    return zk.R1CS(r1.n_public, r1.n_wires, cons)


def test_groth16_setup_and_prove_from_r1cs_on_device(golden):
    """groth16.Setup on the device (explicit toxic waste) + a, b, c = L w, R w, O w + prove: the proof bytes of the committed instances -- made with
    the ORACLE's setup, each verified by pairings -- come out again, i.e. every base point of the device-built key is the oracle's; the verifying
    key's points equal the oracle's too.  Instances: the reference's toy circuit X * Y = Z (main.go:80-107) and a 13-constraint R1CS."""
    from tests.golden.gen_golden import small_r1cs
    toy = ref.R1CS(3, 1, [({3: 1}, {1: 1}, {2: 1})])
    cases = {"toy_x3_y2_z6": (toy, (12345, 111, 222, 333, 444)),
             "seq_r1cs_13": (small_r1cs(0x51, 3, 13)[0], tuple(ref.rand_felts(0x70, 5))),
             "seq_r1cs_13_r0": (small_r1cs(0x51, 3, 13)[0], tuple(ref.rand_felts(0x70, 5)))}
    for e in golden["groth16"]:
        r1, tox = cases[e["name"]]
        w = [h2i(v) for v in e["w"]]
        dev = _oracle_r1cs_to_device(r1)
        a, b, c = dev.eval_abc(mont_limbs(w))
        assert from_mont_limbs(a) == [h2i(v) for v in e["a"]] and from_mont_limbs(b) == [h2i(v) for v in e["b"]] and from_mont_limbs(c) == [h2i(v) for v in e["c"]]
        for tables in (True, False):
            pk, vk = zk.setup(dev, mont_limbs(list(tox)), precompute_tables=tables)
            r, s = mont_limbs([h2i(e["r"])])[0], mont_limbs([h2i(e["s"])])[0]
            assert zk.prove_r1cs(dev, pk, mont_limbs(w), r, s).hex() == e["proof"], e["name"]
            assert zk.prove(pk, a, b, c, mont_limbs(w), r, s).hex() == e["proof"]
            pk.free()
        _, ovk = ref.groth16_setup(r1, *tox)
        assert vk["g1_alpha"].tobytes() == ref.g1_affine_mont_bytes(ovk["g1_alpha"])
        assert [p.tobytes() for p in vk["g1_k"]] == [ref.g1_affine_mont_bytes(p) for p in ovk["g1_ic"]]
        assert (vk["g2_beta"].tobytes(), vk["g2_gamma"].tobytes(), vk["g2_delta"].tobytes()) == tuple(ref.g2_affine_mont_bytes(ovk[k]) for k in ("g2_beta", "g2_gamma", "g2_delta"))
        with pytest.raises(ValueError):
            zk.prove_r1cs(dev, zk.setup(dev, mont_limbs(list(tox)))[0], mont_limbs(w[:-1]), r, s)
        dev.free()


def test_r1cs_sparse_mat_vec_at_2p12_constraints():
    """The solver step on a 2^12-constraint random R1CS (3 + 3 + 2 entries per row): the device's a, b, c = L w, R w, O w equal the oracle's
    big-integer evaluation element for element."""
    g = ref.SplitMix64(0x99)
    nc, nw, npub = 1 << 12, 3000, 4
    w = [1] + [g.felt() for _ in range(nw - 1)]
    cons = []
    for _ in range(nc):
        L = {int(g.next() % nw): g.felt() for _ in range(3)}
        Rr = {int(g.next() % nw): g.felt() for _ in range(3)}
        O = {int(g.next() % nw): g.felt() for _ in range(2)}
        cons.append((L, Rr, O))
    r1 = ref.R1CS(npub, nw - npub, cons)
    dev = _oracle_r1cs_to_device(r1)
    a, b, c = dev.eval_abc(mont_limbs(w))
    ea, eb, ec = r1.eval_abc(w)
    assert from_mont_limbs(a) == ea and from_mont_limbs(b) == eb and from_mont_limbs(c) == ec
    dev.free()


def test_r1cs_load_of_a_large_system_goes_through_the_pinned_ring():
    """zk_bn254_r1cs_load moves arrays of 1 MB and more through a ring of four pinned 8 MB buffers, filled by up to four threads (ctx.hip h2d_big: a cold
    ProveWithPK uploads 0.2 GB of constraint system beside 0.37 GB of key text).  2^19 + 5 constraints with two entries per row of L (33.5 MB of coefficients:
    five pieces, the last one ragged; 4 MB of indices: one piece, one thread), one per row of R, none in O; coefficients drawn from sixteen field elements so that
    big integers can check every row."""
    g = ref.SplitMix64(0xA7)
    nc, nw = (1 << 19) + 5, 1 << 16
    T = [g.felt() for _ in range(16)]
    Tm = mont_limbs(T)
    w = [1] + [g.felt() for _ in range(nw - 1)]
    rng = np.random.default_rng(0xA7)
    li = rng.integers(0, nw, size=2 * nc, dtype=np.uint32)
    lt = rng.integers(0, 16, size=2 * nc)
    ri = rng.integers(0, nw, size=nc, dtype=np.uint32)
    rt = rng.integers(0, 16, size=nc)
    lptr = (2 * np.arange(nc + 1)).astype(np.uint32)
    rptr = np.arange(nc + 1, dtype=np.uint32)
    optr = np.zeros(nc + 1, dtype=np.uint32)
    lval, rval = np.ascontiguousarray(Tm[lt]), np.ascontiguousarray(Tm[rt])
    none_i, none_v = np.zeros(1, np.uint32), np.zeros((1, 4), np.uint64)
    raw = _lib.R1CS(nc, nw, 1, lptr.ctypes.data, li.ctypes.data, lval.ctypes.data, rptr.ctypes.data, ri.ctypes.data, rval.ctypes.data,
                    optr.ctypes.data, none_i.ctypes.data, none_v.ctypes.data)
    h = C.c_uint64(0)
    _lib.check(_lib.lib().zk_bn254_r1cs_load(C.byref(raw), C.byref(h)))
    dev = zk.R1CS.from_handle(h.value, 1, nw, nc)
    a, b, c = dev.eval_abc(mont_limbs(w))
    assert not c.any()
    want_a = [(T[lt[2 * i]] * w[li[2 * i]] + T[lt[2 * i + 1]] * w[li[2 * i + 1]]) % ref.R for i in range(nc)]
    want_b = [T[rt[i]] * w[ri[i]] % ref.R for i in range(nc)]
    assert from_mont_limbs(a) == want_a and from_mont_limbs(b) == want_b
    dev.free()


def test_groth16_from_raw_r1cs_json():
    """The reference's intended Groth16 payload (RawR1CS JSON, src/gnark_backend_wrapper/groth16/acir_to_r1cs.rs:18-60) through buildR1CS
    (backend/groth16/r1cs.go:9-72, restated in frontend.hip and in oracle/plonk_ref.r1cs_from_raw), Setup and Prove on the device: the proof bytes
    equal the oracle's on the same (toxic waste, r, s) and the oracle's pairing verifier accepts them."""
    import json as js
    from noir_backend_using_gnark_amd import frontend as fe
    from oracle import plonk_ref as pl
    hx = lambda v: "%064x" % (v % ref.R)
    w1, w2 = 7, 11
    w3 = w1 * w2 % ref.R
    w4 = (2 * w3 * w1 + 3 * w2 + 5) % ref.R
    values = [w1, w2, w3, w4, 123456789]
    raw = {"gates": [{"mul_terms": [{"coefficient": hx(1), "multiplicand": 1, "multiplier": 2}], "add_terms": [{"coefficient": hx(-1), "sum": 3}], "constant_term": hx(0)},
                     {"mul_terms": [{"coefficient": hx(2), "multiplicand": 3, "multiplier": 1}, {"coefficient": hx(0), "multiplicand": 5, "multiplier": 5}],
                      "add_terms": [{"coefficient": hx(3), "sum": 2}, {"coefficient": hx(-1), "sum": 4}], "constant_term": hx(5)}],
           "public_inputs": [4, 2], "values": ref.felts_wire(values).hex(), "num_variables": 6, "num_constraints": 2}
    r1, wv = pl.r1cs_from_raw(raw)
    a, b, c = r1.eval_abc(wv)
    assert all((x * y - z) % ref.R == 0 for x, y, z in zip(a, b, c)) and r1.n_public == 3 and len(r1.constraints) == 4
    tox, rs = tuple(ref.rand_felts(0xA0, 5)), tuple(ref.rand_felts(0xA1, 2))
    opk, ovk = ref.groth16_setup(r1, *tox)
    proof = ref.groth16_prove(opk, r1.n_public, a, b, c, wv, *rs)
    assert ref.groth16_verify(ovk, proof, wv[:r1.n_public])
    want = ref.groth16_proof_bytes(*proof)
    dev, d_w = fe.groth16_r1cs_from_raw(js.dumps(raw))
    assert (dev.n_public, dev.n_wires) == (r1.n_public, r1.n_wires)
    assert from_mont_limbs(d_w.to_numpy(np.uint64, (dev.n_wires, 4))) == wv
    pk, vk = zk.setup(dev, mont_limbs(list(tox)))
    assert zk.prove_r1cs(dev, pk, d_w, mont_limbs([rs[0]])[0], mont_limbs([rs[1]])[0]) == want
    assert [p.tobytes() for p in vk["g1_k"]] == [ref.g1_affine_mont_bytes(p) for p in ovk["g1_ic"]]
    for bad in ("{}", '{"gates": [], "values": "zz"}', js.dumps(dict(raw, values=raw["values"][:-2])),
                js.dumps(dict(raw, gates=[{"mul_terms": [{"coefficient": hx(1), "multiplicand": 9, "multiplier": 1}], "add_terms": [], "constant_term": hx(0)}]))):
        with pytest.raises(ValueError):
            fe.groth16_r1cs_from_raw(bad)
    pk.free()
    dev.free()
