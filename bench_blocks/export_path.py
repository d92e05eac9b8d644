"""The reference's calls end to end through libgnark_backend.so (Go's C ABI), one child process per nargo-like command: PLONK (the live path) and Groth16 (the metric's proof system, the intended FFI of backend/groth16/r1cs.go)."""
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403
from .common import ROOT, N_PUBLIC, HBM_PEAK_GBS, R_FR, seed_at


def export_path_block(log_gates=19, warm_calls=10):
    """The reference's live call end to end (gnark_backend_ffi/main.go:24-37,58-78; backend/plonk/plonk.go:13-73; backend/common.go:45-76,127-144) through
    libgnark_backend.so's Go ABI: tools/export_bench.py in three child processes -- circuit text, PlonkPreprocess (fresh process), PlonkProveWithPK cold then
    warm + PlonkVerifyWithVK (another fresh process) -- plus the text front end against the document-tree reader it replaced (tools/lower_bench.cpp)."""
    import shutil
    import subprocess
    import tempfile
    d = tempfile.mkdtemp(prefix="zkmi_export_")
    exe = [sys.executable, os.path.join(ROOT, "tools", "export_bench.py")]
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("ZKMI_TEST_NEW_SRS_SIZE", None)

    def run(*a):
        r = subprocess.run(exe + list(a), capture_output=True, text=True, timeout=900, env=env)
        if r.returncode != 0:
            raise RuntimeError("export_bench %s failed: %s" % (a[0], (r.stdout + r.stderr)[-1500:]))
        return json.loads(r.stdout.strip().splitlines()[-1])
    try:
        blk = {"workload": "synthetic ACIR, 2^%d - 8 arithmetic opcodes + 8 public inputs, reference variable layout, 1,000,000-point SRS (backend/common.go:137)" % log_gates,
               "circuit": run("make", d, str(log_gates))}
        blk["preprocess_process"] = run("preprocess", d)
        blk["prove_process"] = run("prove", d, str(warm_calls))
        blk["verify_process"] = run("verify", d)  # a process that only verifies: no HIP runtime
        pp = blk["prove_process"]
        blk["warm_PlonkProveWithPK_ms"] = pp["warm_PlonkProveWithPK_ms"]
        blk["zk_bn254_plonk_prove_ms"] = pp["zk_bn254_plonk_prove_ms"]
        blk["warm_over_prove"] = pp["warm_over_prove"]
        blk["ok"] = bool(pp["verifies"] == 1 and pp["warm_proof_verifies"] == 1 and pp["wrong_public_input_rejected"] == 1 and blk["preprocess_process"]["verifies"] == 1
                        and blk["verify_process"]["verifies"] == 1)
        try:  # the text front end alone, on this box's cores: the streaming reader against the document-tree reader of rounds 1-3
            lb = os.path.join(d, "lower_bench")
            subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tools", "lower_bench.cpp"), "-lpthread", "-o", lb], timeout=300)
            r = subprocess.run([lb, os.path.join(d, "acir.json"), str(blk["circuit"]["witnesses"])], capture_output=True, text=True, timeout=600)
            blk["acir_reader"] = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:  # no compiler on the box: the block stands without it
            blk["acir_reader"] = {"skipped": str(e)[:200]}
        return blk
    finally:
        shutil.rmtree(d, ignore_errors=True)


def export_path_groth16_block(log_constraints=20, warm_calls=10):
    """The metric's own proof system behind the reference's ABI at the metric's size: the intended Groth16 exports (gnark_backend_ffi/backend/groth16/r1cs.go:74-266,
    declared at src/gnark_backend_wrapper/groth16/mod.rs:14-20) through libgnark_backend.so on a synthetic RawR1CS of 2^log constraints
    (tools/synth_raw_r1cs.py; schema src/gnark_backend_wrapper/groth16/acir_to_r1cs.rs:18-60): tools/export_bench_groth16.py in child processes -- Preprocess
    (fresh process), ProveWithPK cold / second (the key's window tables) / warm + VerifyWithVK (another fresh process), VerifyWithVK alone (a third) -- plus
    the RawR1CS reader alone on this box's cores (tools/raw_lower_bench.cpp)."""
    import shutil
    import subprocess
    import tempfile
    d = tempfile.mkdtemp(prefix="zkmi_export_g16_")
    exe = [sys.executable, os.path.join(ROOT, "tools", "export_bench_groth16.py")]
    env = dict(os.environ, PYTHONPATH=ROOT)

    def run(*a):
        r = subprocess.run(exe + list(a), capture_output=True, text=True, timeout=900, env=env)
        if r.returncode != 0:
            raise RuntimeError("export_bench_groth16 %s failed: %s" % (a[0], (r.stdout + r.stderr)[-1500:]))
        return json.loads(r.stdout.strip().splitlines()[-1])
    try:
        blk = {"workload": "synthetic RawR1CS, 2^%d constraints (2^%d gates with one mul term each, 8 public inputs), full-width witness values" % (log_constraints, log_constraints - 1),
               "circuit": run("make", d, str(log_constraints))}
        blk["preprocess_process"] = run("preprocess", d)
        blk["prove_process"] = run("prove", d, str(warm_calls))
        blk["verify_process"] = run("verify", d)
        pp = blk["prove_process"]
        wp = pp["warm_phases_per_call"]
        prove_r1cs = round(wp.get("r1cs_solve_abc", 0.0) + wp.get("groth16_prove", 0.0), 3)  # = zk_bn254_groth16_prove_r1cs of the same instance: a, b, c = L w, R w, O w + the prover
        blk["cold_ProveWithPK_ms"] = pp["cold_ProveWithPK_ms"]
        blk["warm_ProveWithPK_ms"] = pp["warm_ProveWithPK_ms"]
        blk["zk_bn254_groth16_prove_r1cs_ms"] = prove_r1cs
        blk["warm_over_prove"] = round(pp["warm_ProveWithPK_ms"] / prove_r1cs, 3) if prove_r1cs else None
        blk["ok"] = bool(pp["verifies"] == 1 and pp["warm_proof_verifies"] == 1 and pp["wrong_public_input_rejected"] == 1 and blk["preprocess_process"]["verifies"] == 1
                        and blk["verify_process"]["verifies"] == 1)
        try:
            lb = os.path.join(d, "raw_lower_bench")
            subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tools", "raw_lower_bench.cpp"), "-lpthread", "-o", lb], timeout=300)
            r = subprocess.run([lb, os.path.join(d, "raw.json")], capture_output=True, text=True, timeout=600)
            blk["raw_r1cs_reader"] = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:  # no compiler on the box: the block stands without it
            blk["raw_r1cs_reader"] = {"skipped": str(e)[:200]}
        return blk
    finally:
        shutil.rmtree(d, ignore_errors=True)
