"""GPU suite: libgnark_backend.so -- the reference's Go exports under their own names and Go's C ABI (GoString by value, C.CString results, struct
results, one-byte bools), over libzkmi.  Each scenario runs in a subprocess (tests/goffi_worker.py): the shim keeps one SRS per process, creates
<config dir>/noir-lang/srs.hex when missing (backend/common.go:127-144) and ends the process on errors like log.Fatal."""
import json
import os
import subprocess
import sys

import pytest

from oracle import bn254_ref as ref
from oracle import plonk_ref as pl
from tests.helpers import h2i

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def run_worker(tmp_path, job, name, env_extra=None, expect_fail=False):
    f = tmp_path / (name + ".json")
    f.write_text(json.dumps(job))
    env = dict(os.environ, XDG_CONFIG_HOME=str(tmp_path / "cfg"), ZKMI_SRS_SIZE="64", PYTHONPATH=ROOT)
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(HERE, "goffi_worker.py"), str(f)], capture_output=True, text=True, timeout=600, env=env)
    if expect_fail:
        return out
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_plonk_exports_end_to_end_and_the_srs_file(tmp_path):
    """PlonkPreprocess -> PlonkProveWithPK -> PlonkVerifyWithVK on the reference's demo circuits (main.go:223-248) through Go's ABI: the proof verifies,
    a wrong public input does not, PlonkVerifyWithMeta is upstream's `false`; srs.hex appears in the config dir in kzg.SRS.WriteTo's format and a SECOND
    process that finds it there accepts the first process's proof with the first process's key; the oracle's verifier accepts it too."""
    os.makedirs(tmp_path / "cfg", exist_ok=True)
    for k, e in enumerate(json.load(open(os.path.join(HERE, "golden", "plonk_golden.json")))):
        values = [h2i(v) for v in e["values"]]
        wrong = list(values)
        pub_idx = e["acir"]["public_inputs"]
        if pub_idx:
            wrong[pub_idx[0] - 1] = (wrong[pub_idx[0] - 1] + 1) % ref.R
        job = dict(what="plonk", acir=json.dumps(e["acir"]), values=ref.felts_wire(values).hex(), values_wrong_public=ref.felts_wire(wrong).hex(),
                   random_values=ref.felts_wire(ref.rand_felts(77 + k, len(values))).hex())
        a = run_worker(tmp_path, job, "plonk%d" % k)
        assert a["verifies"] == 1 and a["verify_with_meta"] == 0 and len(a["proof"]) == 2 * 548 and len(a["proof_with_meta"]) == 2 * 548
        assert a["verifies_wrong_public"] == (0 if pub_idx else 1), e["name"]
        srs_file = tmp_path / "cfg" / "noir-lang" / "srs.hex"
        assert srs_file.exists()
        srs = pl.kzg_srs_from_bytes(bytes.fromhex(srs_file.read_text()))
        assert len(srs["g1"]) == 64 and srs["g1"][0] == ref.G1_GEN and srs["g2"][0] == ref.G2_GEN
        # second process: loads the file, verifies the first one's proof with the first one's key; a fresh proof with that key verifies too
        b = run_worker(tmp_path, dict(job, pk=a["pk"], vk=a["vk"], proof=a["proof"]), "plonk%d_again" % k)
        assert b["verifies"] == 1 and b["verifies_wrong_public"] == (0 if pub_idx else 1)
        # the oracle: same SRS, its own Setup gives the same key bytes; its verifier accepts the shim's proof
        spr, sol = pl.sparse_r1cs_from_acir(e["acir"], values)
        opk, ovk = pl.plonk_setup(spr, srs)
        assert pl.plonk_pk_bytes(opk).hex() == a["pk"] and pl.plonk_vk_bytes(ovk).hex() == a["vk"]
        p2 = bytes.fromhex(a["proof"])
        pts = [pl.g1_decompress(p2[32 * i:32 * i + 32]) for i in range(7)]
        pr = dict(lro=pts[0:3], z=pts[3], h=pts[4:7], batch_h=pl.g1_decompress(p2[224:256]), claimed=[int.from_bytes(p2[260 + 32 * i:292 + 32 * i], "big") for i in range(7)],
                  z_open_h=pl.g1_decompress(p2[484:516]), zu=int.from_bytes(p2[516:548], "big"))
        assert pl.plonk_verify(ovk, pr, sol[:spr.n_public])


def test_groth16_exports_end_to_end(tmp_path):
    """Preprocess -> ProveWithPK -> VerifyWithVK on a RawR1CS payload through Go's ABI; the oracle's pairing verifier accepts the proof against the
    key image the shim returned; another public input and a tampered proof are rejected."""
    hx = lambda v: "%064x" % (v % ref.R)

    def raw(w4_delta=0):
        w1, w2 = 7, 11
        w3 = w1 * w2 % ref.R
        w4 = (2 * w3 * w1 + 3 * w2 + 5 + w4_delta) % ref.R
        return {"gates": [{"mul_terms": [{"coefficient": hx(1), "multiplicand": 1, "multiplier": 2}], "add_terms": [{"coefficient": hx(-1), "sum": 3}], "constant_term": hx(0)},
                          {"mul_terms": [{"coefficient": hx(2), "multiplicand": 3, "multiplier": 1}], "add_terms": [{"coefficient": hx(3), "sum": 2}, {"coefficient": hx(-1), "sum": 4}],
                           "constant_term": hx(5)}],
                "public_inputs": [4, 2], "values": ref.felts_wire([w1, w2, w3, w4, 99]).hex(), "num_variables": 6, "num_constraints": 2}

    a = run_worker(tmp_path, dict(what="groth16", raw=json.dumps(raw()), raw_other_public=json.dumps(raw(1))), "g16")
    assert a["verifies"] == 1 and a["verifies_other_public"] == 0 and a["verify_with_meta"] == 0 and len(a["proof"]) == 256 and len(a["proof_with_meta"]) == 256
    assert a["verifies_tampered"] == 0
    # the oracle reads the key image and verifies
    vkb = bytes.fromhex(a["vk"])
    nk = int.from_bytes(vkb[288:292], "big")
    ovk = dict(g1_alpha=pl.g1_decompress(vkb[0:32]), g2_beta=pl.g2_decompress(vkb[64:128]), g2_gamma=pl.g2_decompress(vkb[128:192]), g2_delta=pl.g2_decompress(vkb[224:288]),
               g1_ic=[pl.g1_decompress(vkb[292 + 32 * i:324 + 32 * i]) for i in range(nk)])
    r1, wv = pl.r1cs_from_raw(raw())
    pb = bytes.fromhex(a["proof"])
    assert ref.groth16_verify(ovk, (pl.g1_decompress(pb[:32]), pl.g2_decompress(pb[32:96]), pl.g1_decompress(pb[96:])), wv[:r1.n_public])
    opk = pl.groth16_pk_from_bytes(bytes.fromhex(a["pk"]))  # ProvingKey.ReadFrom accepts the image
    assert len(opk["infinity_a"]) == r1.n_wires and len(opk["g1_k"]) == r1.n_wires - r1.n_public


def test_a_corrupt_srs_file_is_replaced(tmp_path):
    """LoadSRS failing for any reason means `generate and save` upstream (common.go:128-141): a truncated srs.hex is overwritten by a fresh SRS and the
    call succeeds."""
    e = json.load(open(os.path.join(HERE, "golden", "plonk_golden.json")))[1]
    cfg = tmp_path / "cfg" / "noir-lang"
    os.makedirs(cfg, exist_ok=True)
    (cfg / "srs.hex").write_text("00ff" * 40)
    values = [h2i(v) for v in e["values"]]
    job = dict(what="plonk", acir=json.dumps(e["acir"]), values=ref.felts_wire(values).hex(), values_wrong_public=ref.felts_wire(values).hex(),
               random_values=ref.felts_wire(values).hex())
    a = run_worker(tmp_path, job, "corrupt")
    assert a["verifies"] == 1
    assert len((cfg / "srs.hex").read_text()) == 2 * (132 + 32 * 64)


def test_errors_end_the_process_like_log_fatal(tmp_path):
    e = json.load(open(os.path.join(HERE, "golden", "plonk_golden.json")))[0]
    out = run_worker(tmp_path, dict(what="fatal", acir=json.dumps(e["acir"]), values=ref.felts_wire([h2i(v) for v in e["values"]]).hex()), "fatal", expect_fail=True)
    assert out.returncode == 1 and "PlonkProveWithPK" in out.stderr
