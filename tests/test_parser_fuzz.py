"""CPU suite: the sanitizer + mutation run of the product's host-side parsers of untrusted text (tests/cpp/parser_fuzz.cpp, built by tests/cpp/Makefile
with -fsanitize=address,undefined from csrc/acir_host.hpp and csrc/text_host.hpp).  The reference ends the process cleanly on every parse failure
(gnark_backend_ffi/main.go:26-30,46-50,61-72: log.Fatal) -- never undefined behaviour -- so: 10^5 deterministic mutants of the reference's three ACIR
fixtures (main.go:233-246), a RawR1CS payload, felt vectors and the golden key / SRS images, every one either accepted or rejected with a status code,
and the streaming front end agrees with the document-tree reader it replaced (tests/cpp/json_dom_ref.hpp) on every accepted output word."""
import json
import os
import subprocess
import sys
import time


from noir_backend_using_gnark_amd import frontend as fe
from oracle import bn254_ref as ref
from oracle import plonk_ref as pl
from tests.helpers import h2i

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from tools import synth_acir  # noqa: E402
from tools import synth_raw_r1cs  # noqa: E402


def _seeds(tmp_path):
    files = []

    def put(name, data):
        f = tmp_path / name
        f.write_bytes(data if isinstance(data, bytes) else data.encode())
        files.append(str(f))

    for k, e in enumerate(json.load(open(os.path.join(HERE, "golden", "plonk_golden.json")))):
        put("s%d.acir.json" % k, json.dumps(e["acir"]))
        put("c%d.acir.json" % k, json.dumps(e["acir"], separators=(",", ":")))
        put("s%d.felts.hex" % k, ref.felts_wire([h2i(v) for v in e["values"]]).hex())
        put("pk%d.bin" % k, bytes.fromhex(e["pk_hex"]))
    for k, e in enumerate(json.load(open(os.path.join(HERE, "golden", "plonk_multi_public_golden.json")))):
        put("m%d.acir.json" % k, json.dumps(e["acir"]))
    acir, _ = synth_acir.synth(40, 3)
    put("synth.acir.json", acir)
    hx = lambda v: "%064x" % (v % ref.R)
    w1, w2 = 7, 11
    w3 = w1 * w2 % ref.R
    w4 = (2 * w3 * w1 + 3 * w2 + 5) % ref.R
    put("g.raw.json", json.dumps({"gates": [
        {"mul_terms": [{"coefficient": hx(1), "multiplicand": 1, "multiplier": 2}], "add_terms": [{"coefficient": hx(-1), "sum": 3}], "constant_term": hx(0)},
        {"mul_terms": [{"coefficient": hx(2), "multiplicand": 3, "multiplier": 1}], "add_terms": [{"coefficient": hx(3), "sum": 2}, {"coefficient": hx(-1), "sum": 4}], "constant_term": hx(5)}],
        "public_inputs": [4, 2], "values": ref.felts_wire([w1, w2, w3, w4, 99]).hex(), "num_variables": 6, "num_constraints": 2}))
    put("synth.raw.json", synth_raw_r1cs.synth(24, 3, seed=9)[0])  # enough gates for the gates array to be split among several readers
    for k, e in enumerate(json.load(open(os.path.join(HERE, "golden", "groth16_wire_golden.json")))[:2]):
        put("g16pk%d.bin" % k, bytes.fromhex(e["pk_hex"]))
    put("srs.bin", pl.kzg_srs_bytes(pl.kzg_new_srs(8, 12345)))
    return files


def test_sanitizer_mutation_run_of_the_host_parsers(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "cpp"), "fuzz"])
    t0 = time.time()
    out = subprocess.run([os.path.join(HERE, "cpp", "build", "parser_fuzz"), "100000"] + _seeds(tmp_path), capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    assert rep["cases"] == 100000 and rep["mismatches"] == 0 and rep["accepted_acir"] > 500, rep  # a mutation run that accepts nothing tests nothing
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]
    assert time.time() - t0 < 150  # 18 s on an idle 8-core container; the bound only says the run stays cheap (a loaded container once took > 60)


def test_streaming_lowering_of_a_synthetic_circuit_matches_the_oracle():
    """zk_acir_to_sparse_r1cs on tools/synth_acir.py's circuit (both gate shapes, a Directive every 1024 opcodes, 8 public inputs -> HandleValues'
    |P| copies per witness) against oracle/plonk_ref.sparse_r1cs_from_acir, both layouts; the synthetic witness satisfies every gate."""
    acir, w = synth_acir.synth(2100, 8, seed=5)
    for layout, name in ((fe.LAYOUT_REFERENCE, "reference"), (fe.LAYOUT_ONE_VAR_PER_WITNESS, "one_var")):
        got = fe.acir_to_sparse_r1cs(acir, len(w), layout)
        spr, sol = pl.sparse_r1cs_from_acir(json.loads(acir), w, layout=name)
        assert got["n_public"] == spr.n_public == 8 and got["n_vars"] == spr.n_vars
        assert got["n_vars"] == (8 * len(w) if layout == fe.LAYOUT_REFERENCE else len(w))
        assert [int(x) + 1 for x in got["order"][:8]] == list(range(1, 9))
        assert [w[int(k)] for k in got["order"]] == list(sol)
        cols = {k: pl.mont_np_to_ints(got[k]) for k in ("ql", "qr", "qo", "qm", "qk")}
        assert len(spr.constraints) == 2100 and spr.is_satisfied(sol)
        for i, (ql, qr, qo, qm, qc, xa, xb, xc) in enumerate(spr.constraints):
            assert (cols["ql"][i], cols["qr"][i], cols["qo"][i], cols["qm"][i], cols["qk"][i]) == (ql, qr, qo, qm, qc), i
            assert (int(got["xa"][i]), int(got["xb"][i]), int(got["xc"][i])) == (xa, xb, xc), i


def test_content_key_separates_texts_and_is_stable_across_thread_counts(tmp_path):
    """The export cache's key (acir_host.hpp content_key, through tools/lower_bench): equal texts give equal keys whatever the segmenting, one flipped bit or
    one byte more gives another key."""
    exe = str(tmp_path / "lower_bench")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tools", "lower_bench.cpp"), "-lpthread", "-o", exe])
    acir, w = synth_acir.synth(8000, 2, seed=9)  # > 64 segments of 64 KB: the threaded key; > 2 MB: the opcodes array goes to several parsers (acir_detail::elements_parallel)
    assert len(acir) > (2 << 20)
    # a 32-byte-aligned block of round 4's fixed multiplier mask zeroed all four lanes of a segment (an absorbing state: whatever came before it in the segment
    # was forgotten); the masks are drawn per process now and a step keeps its input state, so two texts that differ only BEFORE such a block differ in their keys
    blk = bytes.fromhex("d1b54a32d192ed03")[::-1] * 4
    poisoned = [acir[:64] + c + acir[65:96] + blk.decode("latin-1") + acir[128:] for c in "ab"]
    files = []
    for k, text in enumerate([acir, acir, acir[:70000] + ("1" if acir[70000] != "1" else "2") + acir[70001:], acir + " "] + poisoned):
        f = tmp_path / ("t%d.json" % k)
        f.write_bytes(text.encode("latin-1"))
        files.append(str(f))
        if k < 4 and k != 1:
            rep = json.loads(subprocess.run([exe, str(f), str(len(w))], capture_output=True, text=True, check=False).stdout.strip().splitlines()[-1])
            assert rep["same_output"] is True  # the default (several parsers) == one parser == the document-tree reader, word for word
    keys = json.loads(subprocess.run([exe, "--keys"] + files, capture_output=True, text=True, check=True).stdout)  # one process: the masks are per process
    assert keys[0] == keys[1] and len({keys[0], keys[2], keys[3]}) == 3 and keys[4] != keys[5]
    again = json.loads(subprocess.run([exe, "--keys", files[0]], capture_output=True, text=True, check=True).stdout)
    assert again[0] != keys[0]  # another process, other masks: a collision cannot be prepared offline
