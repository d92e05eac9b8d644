"""BASELINE.json configs[4]: the standalone 2^26-point G1 MSM and 2^26 Fr NTT (one GPU, and range / block sharded over ranks), and the SRS read at the reference's 1,000,000 points."""
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403
from .common import ROOT, N_PUBLIC, HBM_PEAK_GBS, R_FR, seed_at


def micro_block(L, lib, zk, log_n, cpu_legs=True):
    """BASELINE.json configs[4] on one GPU: standalone G1 MSM and Fr NTT of 2^log_n, inputs resident in HBM.  Each figure is tied to a check that is not
    the same code path run twice: the MSM equals the recombination of two partial MSMs split at an odd position AND the window-table path over the same
    points registered as resident bases; the transform inverts.  (tools/micro_bench.py is the stand-alone version.)"""
    from noir_backend_using_gnark_amd import bn254 as zb
    MONT = zk.MultiExpConfig(scalars_mont=True)
    n = 1 << log_n
    pts, sc = lib.DeviceBuffer(n * 64), lib.DeviceBuffer(n * 32)
    lib.check(L.zk_bn254_g1_generate_dev(C.c_void_p(pts.ptr), C.c_size_t(n), C.c_uint64(0xB1), None))
    lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(sc.ptr), C.c_size_t(n), C.c_uint64(0xC), C.c_int(1), C.c_int(0), None))
    rb = zb.ResidentBases(pts, n=n)
    r0 = zb.g1_multi_exp_dev(pts.ptr, sc.ptr, n, config=MONT)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        r = zb.g1_multi_exp_dev(pts.ptr, sc.ptr, n, config=MONT)
    dt = (time.perf_counter() - t0) / reps
    m = (n // 3) | 1
    parts = np.stack([zb.g1_multi_exp_dev(pts.ptr, sc.ptr, m, config=MONT, partial=True), zb.g1_multi_exp_dev(pts.ptr + m * 64, sc.ptr + m * 32, n - m, config=MONT, partial=True)])
    t0 = time.perf_counter()
    for _ in range(reps):
        rt = rb.multi_exp_dev(sc, n, config=MONT)
    dt_tab = (time.perf_counter() - t0) / reps
    rb.free()
    out = {"points": n, "g1_msm_ms": round(dt * 1e3, 2), "g1_scalar_muls_per_s": round(n / dt, 1), "g1_msm_hbm_frac": round(96 * n / dt / 8e12, 5),
           "g1_msm_window_tables_ms": round(dt_tab * 1e3, 2), "equals_split_recombination": bool((zb.g1_sum_partials(parts) == r).all() and (r == r0).all()),
           "equals_window_table_path": bool((rt == r).all())}
    sc_h = None
    if cpu_legs:
        # ---- CPU legs (after the MSM's timed loops): the oracle's MultiExp on the SAME 2^log_n points and scalars, downloaded (6 GB at 2^26) -- its point is
        # compared with the GPU's, its time is the reported baseline (plain C, no assembly: not a gnark figure)
        from oracle import oracle as orc
        pts_h, sc_h = pts.to_numpy(np.uint64, (n, 8)), sc.to_numpy(np.uint64, (n, 4))
        t0 = time.perf_counter()
        want = orc.g1_msm(pts_h, sc_h)
        cpu_msm_s = time.perf_counter() - t0
        del pts_h
        out["cpu_baseline_msm"] = {"value": round(n / cpu_msm_s, 1), "unit": "G1 scalar-muls/s", "ms": round(cpu_msm_s * 1e3, 1), "cores": orc.max_threads(), "host_cpus_shown": orc.host_cpus(), "kind": "port",
                                   "sample": "the same 2^%d-point MSM once by oracle/bn254_oracle.c orc_g1_msm (OpenMP bucket method, c = 16 signed digits; plain C, no assembly -- "
                                             "not a gnark figure)" % log_n, "point_matches_gpu": bool((want == r).all())}
    dom = zk.Domain(n)
    head = sc.to_numpy(np.uint64, (4096, 4))
    dom.fft(sc, zk.DIF)
    if cpu_legs:
        gpu_sha = hashlib.sha256(sc.to_numpy(np.uint64, (n, 4)).tobytes()).hexdigest()
        t0 = time.perf_counter()
        orc.fr_ntt(sc_h, False, orc.DIF, inplace=True)
        cpu_ntt_s = time.perf_counter() - t0
        out["cpu_baseline_ntt"] = {"value": round(n / cpu_ntt_s, 1), "unit": "elements/s", "ms": round(cpu_ntt_s * 1e3, 1), "cores": orc.max_threads(), "host_cpus_shown": orc.host_cpus(), "kind": "port",
                                   "sample": "the same 2^%d-point FFT(DIF) once by oracle/bn254_oracle.c orc_fr_ntt (OpenMP radix-2, twiddle table built inside the call; "
                                             "plain C, no assembly -- not a gnark figure)" % log_n,
                                   "image_matches_gpu": bool(hashlib.sha256(sc_h.tobytes()).hexdigest() == gpu_sha)}
        del sc_h
    dom.fft_inverse(sc, zk.DIT)
    out["ntt_inverse_of_forward_is_identity"] = bool((sc.to_numpy(np.uint64, (4096, 4)) == head).all())
    t0 = time.perf_counter()
    for _ in range(5):
        lib.check(L.zk_bn254_ntt_dev(C.c_void_p(sc.ptr), C.c_uint32(log_n), C.c_int(0), C.c_int(zk.DIF), C.c_int(0), C.c_void_p(0)))
    lib.check(L.zk_dev_sync())
    dtn = (time.perf_counter() - t0) / 5
    out.update(ntt_ms=round(dtn * 1e3, 3), ntt_elements_per_s=round(n / dtn, 1), ntt_hbm_frac=round(64 * n / dtn / 8e12, 5))
    pts.free()
    sc.free()
    return out


def micro_sharded_block(L, lib, zk, par, torch, log_total, rank, world):
    """BASELINE.json configs[4] on `world` GPUs: a 2^log_total-point G1 MSM range-sharded over the ranks (each its slice of points and scalars, one all-gather
    of partial sums) and a 2^log_total-point FFT block-sharded over them (parallel.ntt_sharded: two all-to-all transposes per transform).  Checks: every rank
    holds the same MSM result and it equals the recombination of the slices cut at an odd position; FFTInverse(DIT) . FFT(DIF) is the identity on every block."""
    from noir_backend_using_gnark_amd import bn254 as zb
    MONT = zk.MultiExpConfig(scalars_mont=True)
    n = 1 << log_total
    n_loc = n // world
    lo = rank * n_loc
    pts = lib.DeviceBuffer(n_loc * 64)
    lib.check(L.zk_bn254_g1_generate_dev(C.c_void_p(pts.ptr), C.c_size_t(n_loc), C.c_uint64(seed_at(0xB1, 4, lo)), None))
    sc = torch.empty((n_loc, 4), dtype=torch.int64, device="cuda")
    lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(sc.data_ptr()), C.c_size_t(n_loc), C.c_uint64(seed_at(0xC, 4, lo)), C.c_int(1), C.c_int(0), None))
    lib.check(L.zk_dev_sync())

    def sync():
        torch.cuda.synchronize()
        lib.check(L.zk_dev_sync())
        if world > 1 or par._force_collectives():
            par.dist().barrier()

    r0 = par.sharded_g1_multi_exp(pts.ptr, sc.data_ptr(), n_loc, MONT)
    reps = 3
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = par.sharded_g1_multi_exp(pts.ptr, sc.data_ptr(), n_loc, MONT)
    sync()
    dt = (time.perf_counter() - t0) / reps
    m = (n_loc // 3) | 1
    two = np.stack([zb.g1_multi_exp_dev(pts.ptr, sc.data_ptr(), m, config=MONT, partial=True),
                    zb.g1_multi_exp_dev(pts.ptr + m * 64, sc.data_ptr() + m * 32, n_loc - m, config=MONT, partial=True)])
    local = zb.g1_sum_partials(two)                                  # this rank's slice, computed the other way
    one = np.array([0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f], dtype=np.uint64)   # 1 in Montgomery form (Fp)
    rec = np.concatenate([local, one, one]) if local.any() else np.zeros(16, np.uint64)
    recomb = zb.g1_sum_partials(par.all_gather_limbs(rec))
    same = par.all_gather_limbs(r)
    out = {"points": n, "ranks": world, "g1_msm_ms": round(dt * 1e3, 2), "g1_scalar_muls_per_s": round(n / dt, 1),
           "msm_same_on_every_rank": bool((same == same[0]).all() and (r == r0).all()), "msm_equals_odd_split_recombination": bool((recomb == r).all())}
    # FFT(DIF) then FFTInverse(DIT): identity
    x0 = sc.clone()
    y = par.ntt_sharded(sc.clone(), log_total, rank, world, inverse=False, decimation=zk.DIF)
    z = par.ntt_sharded(y, log_total, rank, world, inverse=True, decimation=zk.DIT)
    torch.cuda.synchronize()
    out["ntt_inverse_of_forward_is_identity"] = bool(torch.equal(z, x0))
    work = sc.clone()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        work = par.ntt_sharded(work, log_total, rank, world, inverse=False, decimation=zk.DIF)
    sync()
    dtn = (time.perf_counter() - t0) / reps
    out.update(ntt_ms=round(dtn * 1e3, 3), ntt_elements_per_s=round(n / dtn, 1), ntt_exchanges_per_transform=2 if world > 1 else 0)
    pts.free()
    return out


def srs_block(lib, n=1_000_000):
    """kzg.SRS.ReadFrom of the reference's SRS size (10^6 points, backend/common.go:137) with the G1 points decompressed on the device;
    check: WriteTo(ReadFrom(x)) == x."""
    from noir_backend_using_gnark_amd import kzg
    srs = kzg.new_srs(n, np.array([0x1234567, 0x89abcdef, 0x1111, 0x0222], dtype=np.uint64), table_window_bits=-1)
    raw = srs.write()
    srs.free()
    kzg.read_srs(raw, table_window_bits=-1).free()
    t0 = time.perf_counter()
    s2 = kzg.read_srs(raw, table_window_bits=-1)
    dt = time.perf_counter() - t0
    ok = s2.write() == raw
    s2.free()
    return {"points": n, "read_ms": round(dt * 1e3, 2), "points_per_s": round(n / dt, 1), "bytes": len(raw), "write_of_read_is_identity": bool(ok)}
