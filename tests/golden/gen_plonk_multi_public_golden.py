#!/usr/bin/env python3
"""Generates tests/golden/plonk_multi_public_golden.json: the reference's demo circuit shape (gnark_backend_ffi/main.go:233-246) with TWO and THREE
public inputs, lowered in both variable layouts --

  "reference"  HandleValues exactly as written (/root/reference/gnark_backend_ffi/backend/common.go:45-76): with |P| >= 2 public inputs every
               witness gets one secret variable per non-matching public input and the gates name the last copy.  This is what the reference's
               PlonkPreprocess / PlonkProveWithPK build (its own tests/test_programs/global_consts has `c: pub [Field; ...]`, several public inputs),
               so keys and proofs are only interchangeable with it in this layout.
  "one_var"    one variable per witness (public first) -- the layout without the duplicates.

from the pure-Python restatement of gnark's PLONK backend (oracle/plonk_ref.py).  Reference-derived DATA: the circuit shape and witness literals of
main.go:233-246; everything else (public-input sets, SRS toxic waste, blinders, proofs) is generated here, and every proof is checked by the oracle's
pairing verifier before it is written.  Run:  python tests/golden/gen_plonk_multi_public_golden.py   (deterministic; rewrites the JSON in place)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import bn254_ref as b  # noqa: E402
from oracle import plonk_ref as pl  # noqa: E402
from gen_plonk_golden import M1, ONE, ZERO, acir, hx  # noqa: E402

FIXTURES = [("2_eq_2_public_1_2", acir(ONE, ZERO, [1, 2]), [2, 2, 0, 0, 0, 0]),
            ("0_neq_1_public_2_1", acir(M1, ONE, [2, 1]), [0, 1, b.R - 1, b.R - 1, 1, 0]),      # public inputs listed out of witness order
            ("2_eq_2_public_1_2_5", acir(ONE, ZERO, [1, 2, 5]), [2, 2, 0, 0, 0, 0])]


def main():
    out = []
    for k, (name, circuit, values) in enumerate(FIXTURES):
        alpha = b.rand_felts(0x6125 + k, 1)[0]
        blinders = b.rand_felts(0xB21D + k, 9)
        entry = dict(name=name, acir=circuit, values=[hx(v) for v in values], srs_alpha=hx(alpha), blinders=[hx(v) for v in blinders], layouts={})
        for layout in ("reference", "one_var"):
            spr, sol = pl.sparse_r1cs_from_acir(circuit, values, layout=layout)
            assert spr.is_satisfied(sol), (name, layout)
            n = 1
            while n < len(spr.constraints) + spr.n_public:
                n *= 2
            srs = pl.kzg_new_srs(n + 3, alpha)
            pk, vk = pl.plonk_setup(spr, srs)
            proof = pl.plonk_prove(pk, sol, blinders)
            assert pl.plonk_verify(vk, proof, sol[:spr.n_public]), (name, layout)
            assert not pl.plonk_verify(vk, dict(proof, zu=(proof["zu"] + 1) % b.R), sol[:spr.n_public])
            order, _, _ = pl.handle_values(circuit["public_inputs"], len(values), layout)
            entry["layouts"][layout] = dict(n_public=spr.n_public, n_vars=spr.n_vars, order=[w - 1 for w in order], srs_size=len(srs["g1"]),
                                            gates=[[hx(c) for c in g[:5]] + list(g[5:]) for g in spr.constraints], solution=[hx(v) for v in sol],
                                            vk_hex=pl.plonk_vk_bytes(vk).hex(), pk_hex=pl.plonk_pk_bytes(pk).hex(), proof=pl.plonk_proof_bytes(proof).hex(),
                                            verified_by_pairing=True)
        assert entry["layouts"]["reference"]["pk_hex"] != entry["layouts"]["one_var"]["pk_hex"]  # the point of the fixture
        out.append(entry)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "plonk_multi_public_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
