#!/usr/bin/env python3
"""Generates tests/golden/bn254_golden.json from the pure-Python big-int restatement (oracle/bn254_ref.py).

Nothing here is produced by the reference (Go/Rust, unbuildable in this image: no go/cargo toolchain); the only
reference-derived data are the literals it holds for this path:
  - the "-1 mod r" coefficient of the ACIR fixtures   /root/reference/gnark_backend_ffi/main.go:233
  - the Groth16 toy instance X=3, Y=2, Z=6            /root/reference/gnark_backend_ffi/main.go:81-82,90-107
  - the felt-vector wire layout                       /root/reference/src/gnark_backend_wrapper/serialize.rs:33-47
Run:  python tests/golden/gen_golden.py   (deterministic; rewrites the JSON in place)
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bn254_ref as b  # noqa: E402


def hx(x: int) -> str:
    return "%064x" % x


def sha(vals) -> str:
    """sha256 over the gnark memory image (LE limbs, Montgomery Fr) of a vector of canonical ints."""
    h = hashlib.sha256()
    for v in vals:
        h.update(b.limbs_le(b.to_mont(v, b.R)))
    return h.hexdigest()


def small_r1cs(seed: int, n_public: int, n_constraints: int):
    """Sequential random R1CS: constraint j multiplies two random linear forms of already-known wires and
    defines a fresh secret wire as the product.  Wires: [ONE, public..., secret...]."""
    g = b.SplitMix64(seed)
    n_free = 3
    w = [1] + [g.felt() for _ in range(n_public - 1 + n_free)]
    cons = []
    for _ in range(n_constraints):
        m = len(w)
        L = {int(g.next() % m): g.felt() for _ in range(3)}
        Rr = {int(g.next() % m): g.felt() for _ in range(3)}
        dot = lambda lin: sum(cf * w[i] for i, cf in lin.items()) % b.R
        w.append(dot(L) * dot(Rr) % b.R)
        cons.append((L, Rr, {m: 1}))
    return b.R1CS(n_public, len(w) - n_public, cons), w


def pk_to_json(pk):
    return dict(
        log_domain=pk["domain"].logn,
        g1_alpha=b.g1_affine_mont_bytes(pk["g1_alpha"]).hex(), g1_beta=b.g1_affine_mont_bytes(pk["g1_beta"]).hex(),
        g1_delta=b.g1_affine_mont_bytes(pk["g1_delta"]).hex(),
        g1_a="".join(b.g1_affine_mont_bytes(p).hex() for p in pk["g1_a"]),
        g1_b="".join(b.g1_affine_mont_bytes(p).hex() for p in pk["g1_b"]),
        g1_k="".join(b.g1_affine_mont_bytes(p).hex() for p in pk["g1_k"]),
        g1_z="".join(b.g1_affine_mont_bytes(p).hex() for p in pk["g1_z"]),
        g2_beta=b.g2_affine_mont_bytes(pk["g2_beta"]).hex(), g2_delta=b.g2_affine_mont_bytes(pk["g2_delta"]).hex(),
        g2_b="".join(b.g2_affine_mont_bytes(p).hex() for p in pk["g2_b"]),
    )


def main():
    b._selfcheck()
    out = {}
    # ---- constants / reference literals
    out["constants"] = dict(
        q=hx(b.Q), r=hx(b.R), minus_one_ref_literal="30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000000",
        fr_R=hx(b.MONT_R % b.R), fr_R2=hx(b.MONT_R ** 2 % b.R), fr_ninv="%016x" % ((-b.inv(b.R, 1 << 64)) % (1 << 64)),
        fp_R=hx(b.MONT_R % b.Q), fp_R2=hx(b.MONT_R ** 2 % b.Q), fp_ninv="%016x" % ((-b.inv(b.Q, 1 << 64)) % (1 << 64)),
        root_2_28=hx(b.FR_ROOT_2_28))
    # ---- field KATs (canonical hex in, canonical hex out)
    g = b.SplitMix64(0xF1E1D)
    kats = []
    for m, name in ((b.R, "fr"), (b.Q, "fp")):
        edge = [0, 1, 2, m - 1, m - 2, (m - 1) // 2, b.MONT_R % m]
        vals = edge + [g.felt(m) for _ in range(9)]
        for i in range(len(vals)):
            x, y = vals[i], vals[(i * 7 + 3) % len(vals)]
            kats.append(dict(field=name, a=hx(x), b=hx(y), mul=hx(x * y % m), add=hx((x + y) % m), sub=hx((x - y) % m),
                             inv=hx(b.inv(x, m) if x else 0), a_mont=hx(b.to_mont(x, m))))
    out["field"] = kats
    # ---- wire codec
    felts = [0, 1, b.R - 1, 3]
    out["wire"] = dict(felts=[hx(x) for x in felts], encoded=b.felts_wire(felts).hex())
    # ---- group KATs
    G, H = b.G1_GEN, b.G2_GEN
    ks = [1, 2, 3, 30, b.R - 1, b.R, 0] + b.rand_felts(0x6121, 5)
    out["g1"] = [dict(k=hx(k), mont=b.g1_affine_mont_bytes(b.g1_mul(G, k) if k % b.R else None).hex(),
                      compressed=b.g1_compress(b.g1_mul(G, k) if k % b.R else None).hex()) for k in ks]
    out["g2"] = [dict(k=hx(k), mont=b.g2_affine_mont_bytes(b.g2_mul(H, k) if k % b.R else None).hex(),
                      compressed=b.g2_compress(b.g2_mul(H, k) if k % b.R else None).hex()) for k in ks]
    # ---- MSM KATs: points P_i = k_i*G with k_i = rand_felts(seed_p), scalars = rand_felts(seed_s)
    msm = []
    for (n, sp, ss, kind) in ((4, None, None, "survey_30G"), (1, 21, 22, "uniform"), (33, 23, 24, "uniform"),
                              (64, 25, 26, "uniform"), (64, 27, 28, "edge")):
        if kind == "survey_30G":
            pk_, sc = [1, 2, 3, 4], [1, 2, 3, 4]
        else:
            pk_, sc = b.rand_felts(sp, n), b.rand_felts(ss, n)
            if kind == "edge":  # zeros, ones, r-1, repeated points, small scalars, a point at infinity, 2^k boundaries
                sc[0] = 0; sc[1] = 1; sc[2] = b.R - 1; sc[3] = 1; sc[4] = 0; sc[5] = 2; sc[6] = (1 << 16) - 1; sc[7] = 1 << 15
                sc[8] = (1 << 15) + 1; sc[9] = (1 << 253); sc[10] = b.R - 2; sc[11] = (1 << 128) - 1
                pk_[3] = pk_[1]; pk_[12] = 0; pk_[13] = pk_[14]; sc[13] = (b.R - sc[14]) % b.R  # P13 = P14, s13 = -s14: cancels
        pts1 = [b.g1_mul(G, k) if k % b.R else None for k in pk_]
        pts2 = [b.g2_mul(H, k) if k % b.R else None for k in pk_]
        msm.append(dict(kind=kind, n=n, point_scalars=[hx(k) for k in pk_], scalars=[hx(s) for s in sc],
                        g1=b.g1_affine_mont_bytes(b.msm_naive(b.FP, pts1, sc)).hex(),
                        g2=b.g2_affine_mont_bytes(b.msm_naive(b.FP2, pts2, sc)).hex()))
    out["msm"] = msm
    # ---- NTT KATs
    ntt = [dict(kind="survey_ntt4", n=4, input=[hx(v) for v in (1, 2, 3, 4)],
                natural=[hx(v) for v in b.Domain(4)._ntt_natural([1, 2, 3, 4], b.Domain(4).gen)],
                dif=[hx(v) for v in b.Domain(4).fft([1, 2, 3, 4], b.DIF)])]
    d8 = b.Domain(8); x8 = b.rand_felts(0x88, 8)
    assert d8._ntt_natural(x8, d8.gen) == b.dft_naive(x8, d8.gen)
    for logn, seed in ((0, 30), (1, 31), (3, 0x88), (6, 32), (10, 33), (13, 34)):
        n = 1 << logn
        x = b.rand_felts(seed, n)
        d = b.Domain(n)
        for inverse in (0, 1):
            for dec in (b.DIT, b.DIF):
                for coset in (0, 1):
                    y = (d.fft_inverse if inverse else d.fft)(x, dec, bool(coset))
                    e = dict(kind="modes", log_n=logn, seed=seed, inverse=inverse, decimation=dec, coset=coset, sha256=sha(y))
                    if n <= 8: e["output"] = [hx(v) for v in y]
                    ntt.append(e)
    out["ntt"] = ntt
    # ---- computeH KAT
    ch = []
    for logn, n_used, seed in ((3, 8, 40), (6, 50, 41), (10, 1000, 42)):
        a, bb, c = (b.rand_felts(seed + 100 * i, n_used) for i in range(3))
        h = b.compute_h(a, bb, c, b.Domain(1 << logn))
        ch.append(dict(log_n=logn, n=n_used, seeds=[seed, seed + 100, seed + 200], sha256=sha(h),
                       output=[hx(v) for v in h] if logn <= 3 else None))
    out["compute_h"] = ch
    # ---- Groth16: toy (main.go:80-107) and a sequential random R1CS; explicit toxic waste and (r, s)
    gro = []
    toy = b.R1CS(3, 1, [({3: 1}, {1: 1}, {2: 1})])  # wires [ONE, Y, Z, X]; X*Y = Z
    for name, r1, w, tox, rs in (("toy_x3_y2_z6", toy, [1, 2, 6, 3], (12345, 111, 222, 333, 444), (777, 888)),
                                 ("seq_r1cs_13", *small_r1cs(0x51, 3, 13), tuple(b.rand_felts(0x70, 5)), tuple(b.rand_felts(0x71, 2))),
                                 ("seq_r1cs_13_r0", *small_r1cs(0x51, 3, 13), tuple(b.rand_felts(0x70, 5)), (0, 0))):
        pk, vk = b.groth16_setup(r1, *tox)
        a, bb, c = r1.eval_abc(w)
        assert all((x * y - z) % b.R == 0 for x, y, z in zip(a, bb, c))
        proof = b.groth16_prove(pk, r1.n_public, a, bb, c, w, *rs)
        assert b.groth16_verify(vk, proof, w[:r1.n_public]), name
        gro.append(dict(name=name, n_public=r1.n_public, n_wires=r1.n_wires, pk=pk_to_json(pk),
                        a=[hx(v) for v in a], b=[hx(v) for v in bb], c=[hx(v) for v in c], w=[hx(v) for v in w],
                        r=hx(rs[0]), s=hx(rs[1]), proof=b.groth16_proof_bytes(*proof).hex(), verified_by_pairing=True))
    out["groth16"] = gro
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bn254_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
