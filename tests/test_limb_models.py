"""CPU suite: the bound / exactness models of the 29-bit-limb arithmetic (tools/u29_model.py for the MSM point operations over Fp and
Fp2, tools/u29_ntt_model.py for the NTT butterflies over Fr) must hold, and the constants compiled into csrc/ff29.hpp must be the
ones the models derive."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _run(script):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return out.stdout


def test_msm_point_operation_model_holds():
    out = _run("u29_model.py")
    for needle in ("exact G1 simulation", "exact G2 simulation", "exact G1 add/dbl chain", "exact G2 add/dbl chain", "exact mul: ok",
                   "exact sqr (doubled-operand schedule) == mul(a, a): ok"):
        assert needle in out, needle


def test_ntt_butterfly_model_holds():
    out = _run("u29_ntt_model.py")
    assert out.strip().endswith("OK")
    assert "transforms 2^8" in out and "reduce: exact" in out


def _arrays(struct_name):
    src = open(os.path.join(ROOT, "noir_backend_using_gnark_amd", "csrc", "ff29.hpp")).read()
    body = src[src.index("struct %s {" % struct_name):]
    body = body[:body.index("\n};")]
    arrs = {m.group(1): [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", m.group(2))]
            for m in re.finditer(r"static constexpr uint32_t (\w+)\[9\] = \{([^}]*)\}", body)}
    scal = {m.group(1): int(m.group(2), 16) for m in re.finditer(r"static constexpr uint32_t (\w+) = (0x[0-9a-fA-F]+)u;", body)}
    return arrs, scal


def test_compiled_constants_match_the_models():
    import u29_model as fp
    import u29_ntt_model as fr
    arrs, scal = _arrays("Fp29")
    assert arrs["P"] == fp.PL and scal["NINV"] == fp.NINV
    assert scal["PINV"] == pow(fp.Q, -1, 1 << 29)
    assert arrs["ONE"] == fp.limbs((1 << fp.RBITS) % fp.Q)
    assert arrs["RC"] == fp.RC_P and scal["QM"] == fp.QM_P
    for name, v in arrs.items():
        if name.startswith("BIAS"):
            assert v == fp.bias_limbs(int(name[4:])), name
    arrs, scal = _arrays("Fr29")
    assert arrs["P"] == fr.PL and scal["NINV"] == fr.NINV and arrs["RC"] == fr.RC and scal["QM"] == fr.QM
    for name, v in arrs.items():
        if name.startswith("BIAS"):
            assert v == fr.bias_limbs(int(name[4:])), name
