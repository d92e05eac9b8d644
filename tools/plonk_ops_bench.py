#!/usr/bin/env python3
"""BASELINE.json config 4 at the level of the hot-path operations (the PLONK prover orchestration itself is not part of this
round): the KZG-commit MSMs and coset NTTs that gnark's plonk.Prove issues for a circuit of 2^log_n gates
(~10 commits of size n over the resident SRS, 4 NTTs of size n, ~13 coset NTTs + 1 inverse of size 4n).
Inputs are synthetic and resident in HBM.  usage: plonk_ops_bench.py [log_n=22]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import noir_backend_using_gnark_amd as zk  # noqa: E402
MONT = zk.MultiExpConfig(scalars_mont=True)
from noir_backend_using_gnark_amd import _lib  # noqa: E402
from noir_backend_using_gnark_amd import bn254 as zb  # noqa: E402

L = _lib.lib()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << log_n
srs = _lib.DeviceBuffer((n + 3) * 64)
_lib.check(L.zk_bn254_g1_generate_dev(C.c_void_p(srs.ptr), C.c_size_t(n + 3), C.c_uint64(0x515), None))
polys = [_lib.DeviceBuffer((n + 3) * 32) for _ in range(2)]
for i, p in enumerate(polys):
    _lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(p.ptr), C.c_size_t(n + 3), C.c_uint64(100 + i), C.c_int(1), C.c_int(0), None))
big = _lib.DeviceBuffer(4 * n * 32)
_lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(big.ptr), C.c_size_t(4 * n), C.c_uint64(7), C.c_int(1), C.c_int(0), None))
dom_n, dom_4n = zk.Domain(n), zk.Domain(4 * n)
kzg = zb.ResidentBases(srs, n=n + 3)   # kzg.SRS.G1 resident in HBM with its window tables (built once, like gnark's cached SRS)
# warm up (domain tables, workspaces)
ref_commit = zb.g1_multi_exp_dev(srs.ptr, polys[0].ptr, n + 3, config=MONT)
assert (kzg.multi_exp_dev(polys[0], n + 3, config=MONT) == ref_commit).all(), "table commit != plain commit"
dom_n.fft(polys[1], zk.DIF)
dom_4n.fft(big, zk.DIT, True)
_lib.check(L.zk_dev_sync())

t0 = time.perf_counter()
for i in range(10):
    kzg.multi_exp_dev(polys[i & 1], n + 3, config=MONT)                          # kzg.Commit
t_msm = time.perf_counter() - t0
t0 = time.perf_counter()
for i in range(4):
    (dom_n.fft_inverse if i & 1 else dom_n.fft)(polys[1], zk.DIF)    # n-sized transforms
for i in range(13):
    dom_4n.fft(big, zk.DIT, True)                                    # coset evaluations on 4n
dom_4n.fft_inverse(big, zk.DIF, True)
_lib.check(L.zk_dev_sync())
t_ntt = time.perf_counter() - t0
alg_bytes = 10 * 96 * n + 4 * 64 * n + 14 * 64 * 4 * n
print(json.dumps({"config": "PLONK hot-path ops at 2^%d gates (10 KZG commits, 4 NTT(n), 13 coset NTT(4n) + 1 inverse)" % log_n,
                  "commit_ms_each": round(t_msm / 10 * 1e3, 3), "g1_scalar_muls_per_s": round(10 * (n + 3) / t_msm, 1),
                  "ntt_ms_total": round(t_ntt * 1e3, 3), "ntt_elements_per_s": round((4 * n + 14 * 4 * n) / t_ntt, 1),
                  "ops_ms_total": round((t_msm + t_ntt) * 1e3, 3),
                  "algorithmic_GB": round(alg_bytes / 1e9, 2), "algorithmic_GBps": round(alg_bytes / (t_msm + t_ntt) / 1e9, 1)}))
