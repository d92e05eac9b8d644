#!/usr/bin/env python3
"""BASELINE.json config 5 on one GPU: standalone G1 MSM and Fr NTT of 2^log_n (default 26), inputs resident in HBM.
Reports scalar-muls/s, elements/s and the HBM-roofline fraction of the algorithmic bytes (96 B / scalar-mul, 64 B / element).
Every number is tied to a check that is NOT the same code path run twice: the MSM result must equal the sum of two partial MSMs over a split at
an odd position (other task plan, other bucket loads) AND the window-table path over the same points registered as resident bases; the NTT
must invert (FFTInverse(DIT) . FFT(DIF) = identity on a sampled prefix).  No oracle here (tools/ is product-side): the oracle-backed checks of
the same sizes live in tests/test_gpu_parity.py (test_g1_msm_2p26_properties, test_g1_msm_2p22_vs_oracle)."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import noir_backend_using_gnark_amd as zk  # noqa: E402
from noir_backend_using_gnark_amd import _lib  # noqa: E402
from noir_backend_using_gnark_amd import bn254 as zb  # noqa: E402

MONT = zk.MultiExpConfig(scalars_mont=True)
L = _lib.lib()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << log_n
pts, sc = _lib.DeviceBuffer(n * 64), _lib.DeviceBuffer(n * 32)
t0 = time.perf_counter()
_lib.check(L.zk_bn254_g1_generate_dev(C.c_void_p(pts.ptr), C.c_size_t(n), C.c_uint64(0xB1), None))
t_gen = time.perf_counter() - t0
out = {"log_n": log_n, "generate_points_s": round(t_gen, 2)}
t0 = time.perf_counter()
rb = zb.ResidentBases(pts, n=n)  # kzg-style resident bases: window tables when they fit
out["register_bases_with_tables_s"] = round(time.perf_counter() - t0, 2)
for name, wit in (("uniform", 0), ("witness_like", 1)):
    _lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(sc.ptr), C.c_size_t(n), C.c_uint64(0xC), C.c_int(1), C.c_int(wit), None))
    r0 = zb.g1_multi_exp_dev(pts.ptr, sc.ptr, n, config=MONT)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        r = zb.g1_multi_exp_dev(pts.ptr, sc.ptr, n, config=MONT)
    dt = (time.perf_counter() - t0) / reps
    m = (n // 3) | 1
    parts = np.stack([zb.g1_multi_exp_dev(pts.ptr, sc.ptr, m, config=MONT, partial=True), zb.g1_multi_exp_dev(pts.ptr + m * 64, sc.ptr + m * 32, n - m, config=MONT, partial=True)])
    ok_split = bool((zb.g1_sum_partials(parts) == r).all() and (r == r0).all())
    t0 = time.perf_counter()
    for _ in range(reps):
        rt = rb.multi_exp_dev(sc, n, config=MONT)
    dt_tab = (time.perf_counter() - t0) / reps
    out["g1_msm_" + name] = {"ms": round(dt * 1e3, 2), "scalar_muls_per_s": round(n / dt, 1), "hbm_frac": round(96 * n / dt / 8e12, 5),
                             "window_tables_ms": round(dt_tab * 1e3, 2), "window_tables_scalar_muls_per_s": round(n / dt_tab, 1),
                             "equals_split_recombination": ok_split, "equals_window_table_path": bool((rt == r).all())}
rb.free()
dom = zk.Domain(n)
head = sc.to_numpy(np.uint64, (4096, 4))
dom.fft(sc, zk.DIF)
dom.fft_inverse(sc, zk.DIT)
ok_ntt = bool((sc.to_numpy(np.uint64, (4096, 4)) == head).all())
t0 = time.perf_counter()
for _ in range(5):
    dom.fft(sc, zk.DIF)
_lib.check(L.zk_dev_sync())
dt = (time.perf_counter() - t0) / 5
out["ntt"] = {"ms": round(dt * 1e3, 3), "elements_per_s": round(n / dt, 1), "hbm_frac": round(64 * n / dt / 8e12, 5), "inverse_of_forward_is_identity": ok_ntt}
print(json.dumps(out))
