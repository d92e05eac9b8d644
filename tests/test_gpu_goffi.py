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
    env = dict(os.environ, XDG_CONFIG_HOME=str(tmp_path / "cfg"), ZKMI_TEST_NEW_SRS_SIZE="64", PYTHONPATH=ROOT)  # the worker hands it to zk_export_set_new_srs_size
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


def test_groth16_exports_at_2p11_constraints_and_the_export_worker(tmp_path):
    """The same three exports on tools/synth_raw_r1cs.py's circuit of 2^10 gates = 2^11 constraints (both gate shapes, three public inputs), through
    tools/export_bench_groth16.py -- the worker of bench.py's `export_path_groth16` block -- one process per mode, as nargo would run them: Preprocess (+ a
    ProveWithPK that finds the key it just wrote resident), then cold / second / warm ProveWithPK with the values alternating between two assignments of the
    circuit, VerifyWithVK; then a process that only verifies.  The phases say what was read when: the circuit and the key once, the key's window tables at
    its second proof, nothing but the values afterwards; a verifying process starts no device.  The oracle's pairing verifier accepts the cold proof."""
    exe = [sys.executable, os.path.join(ROOT, "tools", "export_bench_groth16.py")]
    env = dict(os.environ, PYTHONPATH=ROOT)
    d = str(tmp_path)
    run = lambda *a: json.loads(subprocess.run(exe + list(a), capture_output=True, text=True, timeout=900, env=env, check=True).stdout.strip().splitlines()[-1])
    made = run("make", d, "11")
    assert made["constraints"] == 2048 and made["witnesses"] == 1026
    pre = run("preprocess", d)
    assert pre["verifies"] == 1 and "raw_parse_lower" in pre["phases"] and "groth16_setup" in pre["phases"] and "pk_write_hex" in pre["phases"]
    assert "pk_read" not in pre["phases_prove_after_preprocess"] and "raw_parse_lower" not in pre["phases_prove_after_preprocess"]
    pr = run("prove", d, "4")
    assert pr["verifies"] == 1 and pr["warm_proof_verifies"] == 1 and pr["wrong_public_input_rejected"] == 1
    assert pr["resident"]["circuits"] == 1 and pr["resident"]["keys"] == 1
    for k in ("hip_init", "raw_parse_lower", "pk_read", "values_decode", "witness_assemble", "r1cs_solve_abc", "groth16_prove"):
        assert k in pr["cold_phases"], k
    # the key's window tables are built when its SECOND proof is asked for -- on the library's background thread, not inside that call: the second call proves
    # without them (and its proof verifies), the worker then waits for the build (zk_background_wait) and the warm calls find the tables
    # (a background job that finishes before the second call returns is recorded in that call's interval: the two intervals are looked at together)
    after_first = dict(pr["second_phases"], **pr["background_phases"])
    assert "pk_window_tables" not in pr["cold_phases"] and "pk_window_tables" in after_first and "session_streams" in after_first
    assert pr["background_idle"] == 1 and pr["second_proof_verifies"] == 1
    for k in ("raw_parse_lower", "pk_read", "pk_window_tables", "circuit_to_device"):
        assert k not in pr["warm_phases_per_call"] or (k == "circuit_to_device" and pr["warm_phases_per_call"][k] < 0.05), k
    ver = run("verify", d)
    assert ver["verifies"] == 1 and ver["device_entries"] == 0 and "hip_init" not in ver["cold_phases"]
    # ZKMI_TABLE_CAP_GB=0 keeps nothing between calls (the sketch's behaviour: everything read again per call): the proofs still verify
    r0 = subprocess.run(exe + ["prove", d, "2"], capture_output=True, text=True, timeout=900, env=dict(env, ZKMI_TABLE_CAP_GB="0"), check=True)
    p0 = json.loads(r0.stdout.strip().splitlines()[-1])
    assert p0["verifies"] == 1 and p0["warm_proof_verifies"] == 1 and p0["wrong_public_input_rejected"] == 1 and p0["resident"]["circuits"] == 0 and p0["resident"]["keys"] == 0
    assert "raw_parse_lower" in p0["warm_phases_per_call"] and "pk_read" in p0["warm_phases_per_call"]
    # the oracle: the key image reads back, the pairing check accepts the cold proof under the text's public inputs
    raw = json.loads((tmp_path / "raw.json").read_text())
    r1, wv = pl.r1cs_from_raw(raw)
    vkb = bytes.fromhex((tmp_path / "vk.hex").read_text())
    nk = int.from_bytes(vkb[288:292], "big")
    assert nk == r1.n_public == 9
    ovk = dict(g1_alpha=pl.g1_decompress(vkb[0:32]), g2_beta=pl.g2_decompress(vkb[64:128]), g2_gamma=pl.g2_decompress(vkb[128:192]), g2_delta=pl.g2_decompress(vkb[224:288]),
               g1_ic=[pl.g1_decompress(vkb[292 + 32 * i:324 + 32 * i]) for i in range(nk)])
    pb = bytes.fromhex((tmp_path / "proof.hex").read_text())
    assert ref.groth16_verify(ovk, (pl.g1_decompress(pb[:32]), pl.g2_decompress(pb[32:96]), pl.g1_decompress(pb[96:])), wv[:r1.n_public])


def test_the_shim_uses_one_device_unless_asked(tmp_path):
    """libgnark_backend.so starts ONE device entry whatever the node holds (the reference is one process per nargo command; several GPUs in one process have only
    run on virtual entries): the default worker reports 1 entry -- on a multi-GPU box too.  ZKMI_DEVICES opts in: "0,0" (the same GPU listed twice: two virtual
    entries, the SRS spread by range over both, every commitment one partial per entry) still produces a proof the first process's key verifies; a value that is
    neither "all" nor a list of visible ordinals ends the process with a message instead of silently meaning one GPU."""
    e = json.load(open(os.path.join(HERE, "golden", "plonk_golden.json")))[0]
    values = [h2i(v) for v in e["values"]]
    job = dict(what="plonk", acir=json.dumps(e["acir"]), values=ref.felts_wire(values).hex(), values_wrong_public=ref.felts_wire(values).hex(),
               random_values=ref.felts_wire(ref.rand_felts(91, len(values))).hex())
    os.makedirs(tmp_path / "cfg", exist_ok=True)
    a = run_worker(tmp_path, job, "one")
    assert a["verifies"] == 1 and a["device_entries"] == 1
    b = run_worker(tmp_path, dict(job, pk=a["pk"], vk=a["vk"]), "two_virtual", env_extra={"ZKMI_DEVICES": "0,0"})
    assert b["verifies"] == 1 and b["device_entries"] == 2
    c = run_worker(tmp_path, dict(job, pk=a["pk"], vk=a["vk"]), "all", env_extra={"ZKMI_DEVICES": "all"})
    assert c["verifies"] == 1 and c["device_entries"] >= 1
    for bad in ("7,x", "99", "-1", "0,1,2,3,4,5,6,7,8"):
        out = run_worker(tmp_path, dict(job, pk=a["pk"], vk=a["vk"]), "bad", env_extra={"ZKMI_DEVICES": bad}, expect_fail=True)
        assert out.returncode == 1 and "ZKMI_DEVICES" in out.stderr, (bad, out.stderr[-300:])


def test_only_an_unreadable_or_non_hex_srs_file_is_replaced(tmp_path):
    """LoadSRS fails -- and TryLoadSRS generates and saves a new SRS -- exactly when srs.hex cannot be read or is not hex (common.go:86-104, 127-141); what
    ReadFrom makes of hex that is not an SRS is ignored there and the prover then dies on the broken SRS.  Here: a non-hex file is replaced and the call
    succeeds; a file that IS hex but not an SRS ends the process and is left exactly as it was (regenerating would silently invalidate every key issued
    against the old SRS); no temporary file is left behind by the save."""
    e = json.load(open(os.path.join(HERE, "golden", "plonk_golden.json")))[1]
    cfg = tmp_path / "cfg" / "noir-lang"
    os.makedirs(cfg, exist_ok=True)
    values = [h2i(v) for v in e["values"]]
    job = dict(what="plonk", acir=json.dumps(e["acir"]), values=ref.felts_wire(values).hex(), values_wrong_public=ref.felts_wire(values).hex(),
               random_values=ref.felts_wire(values).hex())
    (cfg / "srs.hex").write_text("this is not hex\n")
    a = run_worker(tmp_path, job, "nonhex")
    assert a["verifies"] == 1
    assert len((cfg / "srs.hex").read_text()) == 2 * (132 + 32 * 64)
    assert sorted(p.name for p in cfg.iterdir()) == ["srs.hex", "srs.hex.lock"]
    (cfg / "srs.hex").write_text("00ff" * 40)
    out = run_worker(tmp_path, job, "truncated", expect_fail=True)
    assert out.returncode == 1 and "LoadSRS" in out.stderr
    assert (cfg / "srs.hex").read_text() == "00ff" * 40


def test_concurrent_first_calls_agree_on_one_srs(tmp_path):
    """Two processes that both find no srs.hex (the reference's tests start provers side by side) must end up with ONE SRS: the second waits for the first
    one's file instead of drawing its own alpha, so each accepts the other's proof under the other's key."""
    import threading
    e = json.load(open(os.path.join(HERE, "golden", "plonk_golden.json")))[1]
    os.makedirs(tmp_path / "cfg", exist_ok=True)
    values = [h2i(v) for v in e["values"]]
    job = dict(what="plonk", acir=json.dumps(e["acir"]), values=ref.felts_wire(values).hex(), values_wrong_public=ref.felts_wire(values).hex(),
               random_values=ref.felts_wire(values).hex())
    res = {}
    ts = [threading.Thread(target=lambda k=k: res.__setitem__(k, run_worker(tmp_path, job, "race%d" % k))) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert res[0]["verifies"] == 1 and res[1]["verifies"] == 1
    assert res[0]["vk"] == res[1]["vk"] and res[0]["pk"] == res[1]["pk"]   # same SRS => same Setup output
    c = run_worker(tmp_path, dict(job, pk=res[0]["pk"], vk=res[0]["vk"], proof=res[1]["proof"]), "race_cross")
    assert c["verifies"] == 1


def test_plonk_exports_use_the_reference_variable_layout_for_several_public_inputs(tmp_path):
    """public_inputs = [1, 2] and [1, 2, 5] through Go's ABI: the key bytes PlonkPreprocess returns are those of the oracle's literal HandleValues
    (backend/common.go:45-76) under the SRS the shim wrote -- NOT those of the one-variable-per-witness layout -- and PlonkVerifyWithVK accepts the proof with
    the public values picked in witness order, rejects another public value."""
    os.makedirs(tmp_path / "cfg", exist_ok=True)
    for k, e in enumerate(json.load(open(os.path.join(HERE, "golden", "plonk_multi_public_golden.json")))):
        values = [h2i(v) for v in e["values"]]
        wrong = list(values)
        p0 = e["acir"]["public_inputs"][-1]
        wrong[p0 - 1] = (wrong[p0 - 1] + 1) % ref.R
        job = dict(what="plonk", acir=json.dumps(e["acir"]), values=ref.felts_wire(values).hex(), values_wrong_public=ref.felts_wire(wrong).hex(),
                   random_values=ref.felts_wire(ref.rand_felts(91 + k, len(values))).hex())
        a = run_worker(tmp_path, job, "multi%d" % k)
        assert a["verifies"] == 1 and a["verifies_wrong_public"] == 0, e["name"]
        srs = pl.kzg_srs_from_bytes(bytes.fromhex((tmp_path / "cfg" / "noir-lang" / "srs.hex").read_text()))
        spr, sol = pl.sparse_r1cs_from_acir(e["acir"], values, layout="reference")
        assert (spr.n_public, spr.n_vars) == (e["layouts"]["reference"]["n_public"], e["layouts"]["reference"]["n_vars"])
        opk, ovk = pl.plonk_setup(spr, srs)
        assert pl.plonk_pk_bytes(opk).hex() == a["pk"] and pl.plonk_vk_bytes(ovk).hex() == a["vk"], e["name"]
        spr1, _ = pl.sparse_r1cs_from_acir(e["acir"], values, layout="one_var")
        assert pl.plonk_pk_bytes(pl.plonk_setup(spr1, srs)[0]).hex() != a["pk"]
        p2 = bytes.fromhex(a["proof"])
        pts = [pl.g1_decompress(p2[32 * i:32 * i + 32]) for i in range(7)]
        pr = dict(lro=pts[0:3], z=pts[3], h=pts[4:7], batch_h=pl.g1_decompress(p2[224:256]), claimed=[int.from_bytes(p2[260 + 32 * i:292 + 32 * i], "big") for i in range(7)],
                  z_open_h=pl.g1_decompress(p2[484:516]), zu=int.from_bytes(p2[516:548], "big"))
        assert pl.plonk_verify(ovk, pr, sol[:spr.n_public])


def test_errors_end_the_process_like_log_fatal(tmp_path):
    e = json.load(open(os.path.join(HERE, "golden", "plonk_golden.json")))[0]
    out = run_worker(tmp_path, dict(what="fatal", acir=json.dumps(e["acir"]), values=ref.felts_wire([h2i(v) for v in e["values"]]).hex()), "fatal", expect_fail=True)
    assert out.returncode == 1 and "PlonkProveWithPK" in out.stderr


def test_export_path_worker_cold_and_warm_calls_at_2p10(tmp_path):
    """tools/export_bench.py (the worker of bench.py's `export_path` block) at 2^10 - 8 opcodes + 8 public inputs: PlonkPreprocess in one process,
    PlonkProveWithPK cold + warm and PlonkVerifyWithVK in another, both through Go's ABI.  The key text equals the oracle's Setup under the SRS the first
    process wrote; the cold and the warm proofs verify; another public input is rejected; one circuit and one key stay resident."""
    exe = [sys.executable, os.path.join(ROOT, "tools", "export_bench.py")]
    env = dict(os.environ, PYTHONPATH=ROOT, ZKMI_TEST_NEW_SRS_SIZE="2048")
    d = str(tmp_path)
    run = lambda *a: json.loads(subprocess.run(exe + list(a), capture_output=True, text=True, timeout=900, env=env, check=True).stdout.strip().splitlines()[-1])
    made = run("make", d, "10")
    assert made["opcodes"] == 1016 and made["witnesses"] == 1018
    pre = run("preprocess", d)
    assert pre["verifies"] == 1 and "acir_parse_lower_with_coefficients" in pre["phases"] and "plonk_setup" in pre["phases"]
    assert "pk_text_to_device" not in pre["phases_prove_after_preprocess"]      # the key Preprocess made is resident under the text it returned
    pr = run("prove", d, "3")
    assert pr["verifies"] == 1 and pr["warm_proof_verifies"] == 1 and pr["wrong_public_input_rejected"] == 1
    assert pr["resident"]["circuits"] == 1 and pr["resident"]["keys"] == 1
    for k in ("hip_init", "srs_file_read", "hip_start_wait", "srs_decode", "acir_parse_lower", "pk_text_to_device", "pk_coset_forms", "values_decode", "witness_gather", "plonk_prove"):
        assert k in pr["cold_phases"], k
    # the SRS's window tables wait for the second proving call of a process (a process that makes one proof is better off without them)
    # ... and then on the library's background thread: the second call commits without them, the worker waits for the build before its warm calls
    # (a background job that finishes before the second call returns is recorded in that call's interval: the two intervals are looked at together)
    assert "srs_window_tables" not in pr["cold_phases"] and "srs_window_tables" not in pr["warm_phases_per_call"]
    assert "srs_window_tables" in dict(pr["second_phases"], **pr["background_phases"])
    assert pr["background_idle"] == 1 and pr["second_proof_verifies"] == 1
    # a process that only verifies takes the SRS's two G2 points from the file's header on the host: no HIP runtime, no device entry
    ver = run("verify", d)
    assert ver["verifies"] == 1 and ver["hip_runtime_started"] is False and ver["device_entries"] == 0 and "srs_g2_on_host" in ver["cold_phases"]
    assert "acir_parse_lower" not in pr["warm_phases_per_call"] and "pk_text_to_device" not in pr["warm_phases_per_call"]
    # ZKMI_TABLE_CAP_GB=0 keeps nothing between calls (the reference's behaviour): every call decodes again, the proofs still verify
    env0 = dict(env, ZKMI_TABLE_CAP_GB="0")
    r0 = subprocess.run(exe + ["prove", d, "2"], capture_output=True, text=True, timeout=900, env=env0, check=True)
    p0 = json.loads(r0.stdout.strip().splitlines()[-1])
    assert p0["verifies"] == 1 and p0["warm_proof_verifies"] == 1 and p0["resident"]["circuits"] == 0 and p0["resident"]["keys"] == 0
    assert "pk_text_to_device" in p0["warm_phases_per_call"] and "acir_parse_lower" in p0["warm_phases_per_call"]
    # the oracle's Setup under the same SRS gives the same key text
    srs = pl.kzg_srs_from_bytes(bytes.fromhex(open(os.path.join(d, "cfg", "noir-lang", "srs.hex")).read()))
    acir = json.load(open(os.path.join(d, "acir.json")))
    vh = open(os.path.join(d, "values.hex")).read()
    values = [int(vh[8 + 64 * i:72 + 64 * i], 16) for i in range(int(vh[:8], 16))]
    spr, sol = pl.sparse_r1cs_from_acir(acir, values)
    assert spr.is_satisfied(sol) and spr.n_public == 8 and spr.n_vars == 8 * len(values)
    opk, ovk = pl.plonk_setup(spr, srs, fast=True)
    assert pl.plonk_pk_bytes(opk).hex() == open(os.path.join(d, "pk.hex")).read()


def test_proofs_during_the_background_table_build_in_a_lean_process(tmp_path):
    """A lean process (the export shim's start-up: no high-priority streams), two keys of one circuit, pinned (r, s): every (values, key) pair has ONE right proof.
    Thread A makes key X's FIRST proof while thread B asks for key Y's SECOND one -- which queues Y's window tables and the session's high-priority streams on
    the background thread.  Whatever the interleaving (hi() flipping under a proof in flight: ADVICE r5; a proof racing the build of its own key's tables; a proof
    arriving while one slot is held for its stream), each proof equals the one made sequentially in a fresh state, before and after the tables exist."""
    code = r"""
import ctypes as C, json, sys, threading
sys.path.insert(0, %r)
from noir_backend_using_gnark_amd import _lib
L = _lib.lib()
assert L.zk_init_flags(C.c_uint32(1)) == 0          # ZK_INIT_LEAN_STREAMS, before anything touches a device
from noir_backend_using_gnark_amd import frontend as fe
from oracle import bn254_ref as ref
from tests.helpers import mont_limbs
from tools import synth_raw_r1cs as sr
raw, w = sr.synth(1 << 13, 3, seed=0x91)
raw2, w2 = sr.synth(1 << 13, 3, seed=0x91, first=(0x3333, 0x4444))
rs = mont_limbs(list(ref.rand_felts(0xE1, 2)))
keys = [fe.groth16_preprocess(raw, mont_limbs(list(ref.rand_felts(0xE2 + k, 5)))) for k in range(2)]
texts = [raw, raw2]
want = {}
for t in range(2):
    for k in range(2):
        fe.export_cache_clear()                       # every reference proof is a key's FIRST proof: no tables, no high-priority streams
        want[(t, k)] = fe.groth16_prove_with_pk(texts[t], keys[k][0], rs)
bad = []
for rnd in range(3):
    fe.export_cache_clear()
    assert fe.groth16_prove_with_pk(texts[0], keys[1][0], rs) == want[(0, 1)]   # Y's first proof
    def a():
        for t in (0, 1, 0):
            if fe.groth16_prove_with_pk(texts[t], keys[0][0], rs) != want[(t, 0)]: bad.append(("A", rnd, t))
    def b():
        for t in (1, 0, 1, 0):
            if fe.groth16_prove_with_pk(texts[t], keys[1][0], rs) != want[(t, 1)]: bad.append(("B", rnd, t))
    ths = [threading.Thread(target=a), threading.Thread(target=b)]
    [th.start() for th in ths]
    [th.join() for th in ths]
    assert L.zk_background_wait(C.c_int(-1)) == 1
    for (t, k), p in want.items():                    # with the tables and the streams in place: the same bytes
        if fe.groth16_prove_with_pk(texts[t], keys[k][0], rs) != p: bad.append(("after", rnd, t, k))
info = fe.export_cache_info()
fe.export_cache_clear()
print(json.dumps({"bad": bad, "resident_keys": info["keys"]}))
""" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, PYTHONPATH=ROOT))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["bad"] == [] and d["resident_keys"] == 2


def test_a_process_may_exit_while_its_background_jobs_run(tmp_path):
    """The second proof of a key queues its window tables and the session's streams on the background thread; a process that exits right after that call (nothing
    waits for the build) ends with status 0 and without hanging: exit() cancels the queue, the job in flight stops at its next check, the thread is joined before
    the HIP runtime's own exit handlers run.  Once with the default yield (the job is probably still waiting for the call) and once with none (it is running)."""
    code = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
from noir_backend_using_gnark_amd import _lib, frontend as fe
from oracle import bn254_ref as ref
from tests.helpers import mont_limbs
from tools import synth_raw_r1cs as sr
L = _lib.lib()
assert L.zk_background_set_yield_ms(C.c_int(int(sys.argv[1]))) == 0
raw, w = sr.synth(1 << 14, 3, seed=0x92)
rs = mont_limbs(list(ref.rand_felts(0xE1, 2)))
pk_hex, vk_hex = fe.groth16_preprocess(raw, mont_limbs(list(ref.rand_felts(0xE2, 5))))
a = fe.groth16_prove_with_pk(raw, pk_hex, rs)
b = fe.groth16_prove_with_pk(raw, pk_hex, rs)   # the key's second proof: tables and streams are queued
assert a == b
print("done", L.zk_background_wait(C.c_int(0)))
""" % ROOT
    for yield_ms in ("250", "0"):
        out = subprocess.run([sys.executable, "-c", code, yield_ms], capture_output=True, text=True, timeout=300, env=dict(os.environ, PYTHONPATH=ROOT))
        assert out.returncode == 0 and "done" in out.stdout, (yield_ms, out.stdout[-1000:], out.stderr[-3000:])
