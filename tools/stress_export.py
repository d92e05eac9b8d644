#!/usr/bin/env python3
"""Stability of the export path's warm calls (frontend.hip): two value vectors of one circuit x two proving keys of the SAME text length, pinned (r, s) --
every combination has one right answer.  The calls come in an order that makes the warm path guess right (same pair as the last call), guess the wrong key,
and guess while another thread is proving; every proof is compared with the first one made for its combination and each combination is verified on the host
(under its own key only).

    python tools/stress_export.py [log_gates] [calls]

Prints one JSON object; exit status 1 on any mismatch."""
import json
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from noir_backend_using_gnark_amd import _lib, frontend as fe, verify as vf  # noqa: E402
from oracle import bn254_ref as ref  # noqa: E402
from tests.helpers import mont_limbs  # noqa: E402
from tools import synth_raw_r1cs as sr  # noqa: E402

log_g = int(sys.argv[1]) if len(sys.argv) > 1 else 12
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 400
fe.export_cache_clear()
raw, w = sr.synth(1 << log_g, 3, seed=0x77)
raw2, w2 = sr.synth(1 << log_g, 3, seed=0x77, first=(0x1111, 0x2222))
rs = mont_limbs(list(ref.rand_felts(0xD1, 2)))
keys = [fe.groth16_preprocess(raw, mont_limbs(list(ref.rand_felts(0xD2 + k, 5)))) for k in range(2)]
assert len(keys[0][0]) == len(keys[1][0]) and keys[0][0] != keys[1][0]
texts = [raw, raw2]
pubs = [fe.groth16_public_inputs(t) for t in texts]
first, bad, lock = {}, [], threading.Lock()


def one(i):
    t, k = (i >> 1) & 1, (i // 3) & 1  # runs of equal pairs, key flips inside them
    p = fe.groth16_prove_with_pk(texts[t], keys[k][0], rs)
    with lock:
        if (t, k) not in first:
            first[(t, k)] = p
        elif first[(t, k)] != p:
            bad.append((i, t, k))


for i in range(calls // 2):
    one(i)
ths = [threading.Thread(target=lambda lo: [one(i) for i in range(lo, calls // 2 + lo)], args=(lo,)) for lo in (0, 1)]
for th in ths:
    th.start()
for th in ths:
    th.join()
ok = all(vf.groth16_verify(bytes.fromhex(p), keys[k][1], pubs[t]) for (t, k), p in first.items())
cross = any(vf.groth16_verify(bytes.fromhex(p), keys[1 - k][1], pubs[t]) for (t, k), p in first.items())
out = {"log_gates": log_g, "groth16_calls": calls // 2 * 3, "combinations": len(first), "mismatches": len(bad), "all_verify": bool(ok), "verify_under_the_other_key": bool(cross),
       "resident": fe.export_cache_info()}
fe.export_cache_clear()
print(json.dumps(out))
sys.exit(0 if (not bad and ok and not cross and len(first) == 4) else 1)
