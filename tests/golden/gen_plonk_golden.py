#!/usr/bin/env python3
"""Generates tests/golden/plonk_golden.json from the pure-Python restatement of gnark's PLONK backend (oracle/plonk_ref.py).

Reference-derived DATA (inputs only; nothing here is produced by running the reference, which cannot be built in this image):
  - the three ACIR circuits and witness vectors of the reference's own PLONK demo      /root/reference/gnark_backend_ffi/main.go:223-248
    ("0 != 1", "2 == 2", "3 == 3 (no public)"; witness literals {0,1,-1,-1,1,0}, {2,2,0,0,0,0}, {3,3,0,0,0,0})
  - their lowering to gates                                                             backend/plonk/sparse_r1cs.go:44-107
Everything else (SRS toxic waste, blinders, proofs) is generated here; every proof is checked by the oracle's pairing verifier before
it is written.  Run:  python tests/golden/gen_plonk_golden.py   (deterministic; rewrites the JSON in place)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bn254_ref as b  # noqa: E402
from oracle import plonk_ref as pl  # noqa: E402

M1 = "30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000000"  # -1 mod r, the literal of main.go:233
ONE = "0" * 63 + "1"
ZERO = "0" * 64


def arith(mul, lin, qc):
    return {"Arithmetic": {"mul_terms": mul, "linear_combinations": lin, "q_c": qc}}


def acir(last_lin, last_qc, public):
    """The circuit shape shared by the three demo fixtures (main.go:233, 239, 245): w1 - w2 - w3 = 0; Invert directive; w3*w4 - w5 = 0;
    w3*w5 - w3 = 0; a final linear gate on w5."""
    return {"current_witness_index": 6,
            "opcodes": [arith([], [[ONE, 1], [M1, 2], [M1, 3]], ZERO),
                        {"Directive": {"Invert": {"x": 3, "result": 4}}},
                        arith([[ONE, 3, 4]], [[M1, 5]], ZERO),
                        arith([[ONE, 3, 5]], [[M1, 3]], ZERO),
                        arith([], [[last_lin, 5]], last_qc)],
            "public_inputs": public}


FIXTURES = [("0_neq_1", acir(M1, ONE, [2]), [0, 1, b.R - 1, b.R - 1, 1, 0]),
            ("2_eq_2", acir(ONE, ZERO, [2]), [2, 2, 0, 0, 0, 0]),
            ("3_eq_3_no_public", acir(ONE, ZERO, []), [3, 3, 0, 0, 0, 0])]


def hx(x):
    return "%064x" % x


def main():
    out = []
    for k, (name, circuit, values) in enumerate(FIXTURES):
        spr, sol = pl.sparse_r1cs_from_acir(circuit, values)
        assert spr.is_satisfied(sol), name
        alpha = b.rand_felts(0x5125 + k, 1)[0]
        srs = pl.kzg_new_srs(8 + 3, alpha)
        pk, vk = pl.plonk_setup(spr, srs)
        blinders = b.rand_felts(0xB11D + k, 9)
        trace = {}
        proof = pl.plonk_prove(pk, sol, blinders, trace=trace)
        assert pl.plonk_verify(vk, proof, sol[:spr.n_public]), name
        assert not pl.plonk_verify(vk, dict(proof, zu=(proof["zu"] + 1) % b.R), sol[:spr.n_public])
        # the same instance with all five challenges pinned by the caller
        pinned = dict(zip(("gamma", "beta", "alpha", "zeta", "kzg_gamma"), b.rand_felts(0xC4A1 + k, 5)))
        proof_p = pl.plonk_prove(pk, sol, blinders, challenges=pinned)
        assert pl.plonk_verify(vk, proof_p, sol[:spr.n_public], challenges=pinned), name
        out.append(dict(name=name, acir=circuit, values=[hx(v) for v in values], n_public=spr.n_public, n_vars=spr.n_vars,
                        gates=[[hx(c) for c in g[:5]] + list(g[5:]) for g in spr.constraints], solution=[hx(v) for v in sol],
                        srs_alpha=hx(alpha), srs_size=len(srs["g1"]), blinders=[hx(v) for v in blinders],
                        vk={kk: b.g1_affine_mont_bytes(vk[kk]).hex() for kk in ("ql", "qr", "qm", "qo", "qk")} | {"s": [b.g1_affine_mont_bytes(p).hex() for p in vk["s"]]},
                        challenges={kk: hx(trace[kk]) for kk in ("gamma", "beta", "alpha", "zeta", "kzg_gamma")},
                        vk_hex=pl.plonk_vk_bytes(vk).hex(), pk_hex=pl.plonk_pk_bytes(pk).hex(),  # VerifyingKey.WriteTo / ProvingKey.WriteTo images
                        proof=pl.plonk_proof_bytes(proof).hex(), verified_by_pairing=True,
                        pinned_challenges={kk: hx(v) for kk, v in pinned.items()}, proof_pinned=pl.plonk_proof_bytes(proof_p).hex()))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "plonk_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
