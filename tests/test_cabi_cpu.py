"""CPU suite: the C-ABI library loads without a GPU, exports every symbol include/zkmi.h declares, fails loudly (no CPU
fallback) and its host-side arithmetic (Horner / affine conversion / partial-sum combine) agrees with the oracle."""
import os
import re

import numpy as np
import pytest

from noir_backend_using_gnark_amd import _lib
from noir_backend_using_gnark_amd import bn254 as zb
from oracle import bn254_ref as ref
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MONT = zb.MultiExpConfig(scalars_mont=True)


def test_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "zkmi.h")).read()
    declared = set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.lib()
    for s in declared:
        assert hasattr(lib, s), s
    assert b"gfx950" in lib.zk_version()


def test_host_selftest():
    assert _lib.lib().zk_selftest_host() == 0


def test_no_cpu_fallback_without_gpu():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.ZkmiError) as ei:
        zb.g1_multi_exp(np.zeros((2, 8), np.uint64), np.zeros((2, 4), np.uint64), config=MONT)
    assert ei.value.code == _lib.ZK_ERR_NO_DEVICE
    with pytest.raises(_lib.ZkmiError):
        zb.Domain(8).fft(np.zeros((8, 4), np.uint64), zb.DIF)
    # argument validation happens before the device is touched and mirrors upstream's MultiExp errors
    with pytest.raises(ValueError, match=r"len\(points\) != len\(scalars\)"):
        zb.g1_multi_exp(np.zeros((2, 8), np.uint64), np.zeros((3, 4), np.uint64), config=MONT)
    with pytest.raises(ValueError, match="NbTasks"):
        zb.g2_multi_exp(np.zeros((2, 16), np.uint64), np.zeros((2, 4), np.uint64), zb.MultiExpConfig(scalars_mont=True, nb_tasks=1025))


def test_host_partial_sum_combine_matches_oracle():
    """zk_bn254_g1_sum_xyzz / g2 (host-side tail of the range-sharded multi-GPU MSM): XYZZ partials with ZZ=ZZZ=1."""
    one = np.frombuffer(ref.limbs_le(ref.to_mont(1, ref.Q)), dtype=np.uint64)
    pts = orc.g1_gen_points(5, 6)
    parts = np.zeros((7, 16), np.uint64)
    for i in range(6):
        parts[i, :8] = pts[i]
        parts[i, 8:12] = one
        parts[i, 12:16] = one
    exp = pts[0]
    for i in range(1, 6):
        exp = orc.g1_add(exp, pts[i])
    assert (zb.g1_sum_partials(parts) == exp).all()  # row 6 is the point at infinity (ZZ = 0)
    p2 = orc.g2_gen_points(6, 3)
    parts2 = np.zeros((3, 32), np.uint64)
    for i in range(3):
        parts2[i, :16] = p2[i]
        parts2[i, 16:20] = one
        parts2[i, 24:28] = one
    assert (zb.g2_sum_partials(parts2) == orc.g2_add(orc.g2_add(p2[0], p2[1]), p2[2])).all()
    assert (zb.g1_sum_partials(np.zeros((0, 16), np.uint64)) == 0).all()


def test_cpp_host_mirror_compiles_and_keeps_upstream_error_behaviour(tmp_path):
    """include/zkmi.hpp (C++ mirror of the gnark-crypto interface) compiles against the C ABI, and upstream's two MultiExp errors plus
    the domain-size check come back without touching a device."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "mirror_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "mirror_check.cpp"),
                           "-L" + os.path.join(root, "noir_backend_using_gnark_amd"), "-lzkmi", "-Wl,-rpath," + os.path.join(root, "noir_backend_using_gnark_amd"),
                           "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_acir_lowering_matches_the_oracle_on_the_reference_fixtures():
    """zk_acir_to_sparse_r1cs (host code, no GPU needed) against oracle/plonk_ref.sparse_r1cs_from_acir on the reference's three demo circuits
    (gnark_backend_ffi/main.go:223-248; lowering per backend/plonk/sparse_r1cs.go:44-107, variable order per backend/common.go:45-76)."""
    import json
    from noir_backend_using_gnark_amd import frontend as fe
    from oracle import plonk_ref as pl
    for e in json.load(open(os.path.join(ROOT, "tests", "golden", "plonk_golden.json"))):
        values = [int(v, 16) for v in e["values"]]
        spr, sol = pl.sparse_r1cs_from_acir(e["acir"], values)
        got = fe.acir_to_sparse_r1cs(json.dumps(e["acir"]), len(values))
        assert got["n_public"] == spr.n_public and got["n_vars"] == spr.n_vars
        g = spr.constraints
        for k, name in enumerate(("ql", "qr", "qo", "qm", "qk")):
            assert (got[name] == pl.ints_to_mont_np([c[k] for c in g])).all(), name
        assert [list(got[n]) for n in ("xa", "xb", "xc")] == [[c[5] for c in g], [c[6] for c in g], [c[7] for c in g]]
        assert [values[i] for i in got["order"]] == sol
    # malformed inputs are errors, not crashes
    for bad in ("", "[]", '{"opcodes": 3}', '{"opcodes":[{"Foo":{}}],"public_inputs":[]}', '{"opcodes":[{"Arithmetic":{"mul_terms":[["zz",1,2]],"linear_combinations":[],"q_c":"00"}}]}',
                '{"opcodes":[{"Arithmetic":{"mul_terms":[],"linear_combinations":[["01",99]],"q_c":"00"}}],"public_inputs":[]}', '{"opcodes":[' + "[" * 100):
        with pytest.raises(ValueError):
            fe.acir_to_sparse_r1cs(bad, 6)
    # two public inputs: each witness is one variable, the public ones first in witness order
    two = {"current_witness_index": 4, "public_inputs": [3, 1], "opcodes": [{"Arithmetic": {"mul_terms": [["01", 1, 2]], "linear_combinations": [["%064x" % (ref.R - 1), 3]], "q_c": "00"}}]}
    got = fe.acir_to_sparse_r1cs(json.dumps(two), 4)
    assert got["n_public"] == 2 and got["n_vars"] == 4 and list(got["order"]) == [0, 2, 1, 3] and (got["xa"][0], got["xb"][0], got["xc"][0]) == (0, 2, 1)
