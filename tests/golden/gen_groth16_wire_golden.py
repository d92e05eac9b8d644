"""Generates tests/golden/groth16_wire_golden.json: the wire images (hex of ProvingKey.WriteTo / VerifyingKey.WriteTo, gnark v0.8.0 layout as restated
in oracle/plonk_ref.py) of the keys behind the committed Groth16 proofs in bn254_golden.json -- same circuits, same toxic waste -- so that the key
reader / writer of the product can be checked against bytes that are tied to pairing-verified proofs.
Run from the repo root:  python tests/golden/gen_groth16_wire_golden.py"""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import bn254_ref as b  # noqa: E402
from oracle import plonk_ref as pl  # noqa: E402
from tests.golden.gen_golden import small_r1cs  # noqa: E402


def main():
    toy = b.R1CS(3, 1, [({3: 1}, {1: 1}, {2: 1})])  # wires [ONE, Y, Z, X]; X*Y = Z  (reference main.go:80-107)
    out = []
    for name, r1, tox in (("toy_x3_y2_z6", toy, (12345, 111, 222, 333, 444)),
                          ("seq_r1cs_13", small_r1cs(0x51, 3, 13)[0], tuple(b.rand_felts(0x70, 5)))):
        pk, vk = b.groth16_setup(r1, *tox)
        pkb = pl.groth16_pk_bytes(pk)
        back = pl.groth16_pk_from_bytes(pkb)
        assert pl.groth16_pk_bytes(back) == pkb
        assert pl.groth16_pk_expand(back)["g1_a"] == pk["g1_a"] and pl.groth16_pk_expand(back)["g2_b"] == pk["g2_b"]
        vkb = pl.groth16_vk_bytes(dict(vk, g1_beta=pk["g1_beta"], g1_delta=pk["g1_delta"]))
        out.append(dict(name=name, n_wires=r1.n_wires, n_public=r1.n_public, nb_infinity_a=sum(back["infinity_a"]), nb_infinity_b=sum(back["infinity_b"]),
                        pk_hex=pkb.hex(), vk_hex=vkb.hex(), pk_sha256=hashlib.sha256(pkb).hexdigest()))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "groth16_wire_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
