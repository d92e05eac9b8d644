"""north_star's literal flow: the hot operations of one 2^20 proof call by call through the inner C ABI with host slices; and the two-slice recombination check of the 2^24 block."""
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403
from .common import ROOT, N_PUBLIC, HBM_PEAK_GBS, R_FR, seed_at


def two_slice_recombination(inst, par, lib, L):
    """The proof of `inst` recomputed WITHOUT the key's window tables and without the single-call schedule: computeH, then the five
    MSMs of each half of the wires / coefficients through zk_bn254_groth16_msm5_dev (planner's plain window width for 2^(log_n-1)
    points, per-window bucket sets, host Horner), then zk_bn254_groth16_finalize on the two partial records."""
    N, npub = inst.N, inst.n_public
    d_h = lib.DeviceBuffer(N * 32)
    lib.check(L.zk_bn254_groth16_compute_h_dev(C.c_void_p(inst.d_a.ptr), C.c_void_p(inst.d_b.ptr), C.c_void_p(inst.d_c.ptr), C.c_size_t(N),
                                               C.c_uint32(inst.log_n), C.c_void_p(d_h.ptr), None))
    recs = []
    for rank in (0, 1):
        lo, hi = rank * N // 2, (rank + 1) * N // 2
        skip = npub if rank == 0 else 0
        nz = (hi - lo) - (1 if rank == 1 else 0)
        recs.append(par.groth16_msm5_local(inst.g1_a.ptr + lo * 64, inst.g1_b.ptr + lo * 64, inst.g2_b.ptr + lo * 128, inst.d_w.ptr + lo * 32, hi - lo,
                                           inst.g1_k.ptr + (lo + skip) * 64, inst.d_w.ptr + (lo + skip) * 32, hi - lo - skip,
                                           inst.g1_z.ptr + lo * 64, d_h.ptr + lo * 32, nz))
    d_h.free()
    return par.groth16_finalize(inst.pk, np.stack(recs), inst.r, inst.s)


def inner_boundary_block(L, lib, zk, par, inst, proof):
    """north_star's literal flow: gnark itself keeps running groth16.Prove and calls the replacement at its gnark-crypto call sites -- MultiExp x5 and
    (*Domain).FFT / FFTInverse x7 -- through the inner C ABI with HOST slices (zk_bn254_msm_bases against base arrays registered once
    per key, zk_bn254_ntt in place on the caller's slice).  Timed here call by call, scalars / coefficients crossing PCIe inside every call:
      * the seven transforms of computeH in gnark's order: FFTInverse(DIF) x3, FFT(DIT, coset) x3, then FFTInverse(DIF, coset) (on one of the arrays: the
        pointwise step between them is gnark's own Go code and is not part of the boundary);
      * the five MultiExp calls, one after the other and -- as gnark issues them -- from five concurrent host threads.
    Check: the proof assembled from the five affine results (h from the fused computeH entry point) through zk_bn254_groth16_finalize equals the bytes of
    the single-call prover."""
    import threading
    from noir_backend_using_gnark_amd import bn254 as zb, groth16 as zg
    N, npub = inst.N, inst.n_public
    MONT = zk.MultiExpConfig(scalars_mont=True)
    ha, hb, hc, hw = (d.to_numpy(np.uint64, (N, 4)) for d in (inst.d_a, inst.d_b, inst.d_c, inst.d_w))
    t0 = time.perf_counter()
    bases = [zb.ResidentBases(inst.g1_a, n=N), zb.ResidentBases(inst.g1_b, n=N), zb.ResidentBases(inst.g1_k.ptr + npub * 64, n=N - npub),
             zb.ResidentBases(inst.g1_z, n=N), zb.ResidentBases(inst.g2_b, is_g2=True, n=N)]
    lib.check(L.zk_dev_sync())
    reg_s = time.perf_counter() - t0
    dom = zk.Domain(N)
    h = zg.compute_h(ha, hb, hc, inst.log_n)
    scal = [hw, hw, hw[npub:], h[:N - 1], hw]

    wake = np.zeros((1024, 4), np.uint64)
    dom_wake = zk.Domain(1024)

    def ntt7():
        a, b, c = ha.copy(), hb.copy(), hc.copy()
        # the three copies above leave the GPU idle for ~30 ms and on some boxes the first call after such a pause pays 8-20 ms of wake-up (seen with any build):
        # one untimed 1024-point transform first, so that the seven calls are timed as gnark would issue them -- back to back
        dom_wake.fft(wake, zk.DIF)
        ts = []
        for f in ([lambda v=v: dom.fft_inverse(v, zk.DIF) for v in (a, b, c)] + [lambda v=v: dom.fft(v, zk.DIT, True) for v in (a, b, c)] +
                  [lambda: dom.fft_inverse(a, zk.DIF, True)]):
            t = time.perf_counter()
            f()
            ts.append((time.perf_counter() - t) * 1e3)
        return ts

    def msm5_seq():
        dom_wake.fft(wake, zk.DIF)  # as in ntt7: the calls are timed back to back, not after a pause
        out, ts = [], []
        for bs, sc in zip(bases, scal):
            t = time.perf_counter()
            out.append(bs.multi_exp(sc, config=MONT))
            ts.append((time.perf_counter() - t) * 1e3)
        return out, ts

    def msm5_conc():
        out = [None] * 5
        th = [threading.Thread(target=lambda k=k: out.__setitem__(k, bases[k].multi_exp(scal[k], config=MONT))) for k in range(5)]
        dom_wake.fft(wake, zk.DIF)
        t = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        return out, (time.perf_counter() - t) * 1e3

    def ntt7_chains():
        """the same seven transforms as INTEGRATION.md 4 patches computeH to issue them: a, b, c each go through FFTInverse(DIF) then FFT(DIT, coset) on a goroutine of
        their own (the three chains are independent), then the closing FFTInverse(DIF, coset) -- downloads of one chain overlap the uploads of another (PCIe is full duplex)"""
        a, b, c = ha.copy(), hb.copy(), hc.copy()
        dom_wake.fft(wake, zk.DIF)

        def chain(v):
            dom.fft_inverse(v, zk.DIF)
            dom.fft(v, zk.DIT, True)
        th = [threading.Thread(target=chain, args=(v,)) for v in (a, b, c)]
        t = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dom.fft_inverse(a, zk.DIF, True)
        return (time.perf_counter() - t) * 1e3

    def msm5_prepared():
        """the five MultiExp calls as INTEGRATION.md 4 patches groth16.Prove to issue them: the wire values are registered once (zk_bn254_scalars_register: one upload,
        one recoding shared by A, B1, G2.B, a second one for K from the resident copy), h goes through zk_bn254_msm_bases; five concurrent host threads"""
        out = [None] * 5
        dom_wake.fft(wake, zk.DIF)
        t = time.perf_counter()
        S = zb.PreparedScalars(hw, MONT)
        jobs = [lambda: bases[0].multi_exp_prepared(S), lambda: bases[1].multi_exp_prepared(S), lambda: bases[2].multi_exp_prepared(S, skip=npub),
                lambda: bases[3].multi_exp(scal[3], config=MONT), lambda: bases[4].multi_exp_prepared(S)]
        th = [threading.Thread(target=lambda k=k: out.__setitem__(k, jobs[k]())) for k in (3, 4, 0, 1, 2)]  # Z's upload and G2.B (the longest) first
        for x in th:
            x.start()
        for x in th:
            x.join()
        ms = (time.perf_counter() - t) * 1e3
        S.free()
        return out, ms

    def prove_call_pattern():
        """The calls as groth16.Prove overlaps them (gnark v0.8.0 groth16 prove.go: computeH runs on a goroutine of its own beside the MultiExp goroutines; only the
        Z MultiExp waits for h): the seven transforms (three chains + the closing one) and then Z on one thread, A, B1, K, G2.B against the registered wire values
        on four others.  The transforms are PCIe-bound, the MultiExps ALU-bound -- they share the machine.  Wall time of the whole pattern."""
        a, b, c = ha.copy(), hb.copy(), hc.copy()
        out = [None] * 5
        dom_wake.fft(wake, zk.DIF)
        t = time.perf_counter()
        S = zb.PreparedScalars(hw, MONT)

        def h_then_z():
            def chain(v):
                dom.fft_inverse(v, zk.DIF)
                dom.fft(v, zk.DIT, True)
            th = [threading.Thread(target=chain, args=(v,)) for v in (a, b, c)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            dom.fft_inverse(a, zk.DIF, True)
            out[3] = bases[3].multi_exp(scal[3], config=MONT)  # the true h (the pointwise step between the transforms is gnark's Go code, not run here)
        jobs = {3: h_then_z, 4: lambda: out.__setitem__(4, bases[4].multi_exp_prepared(S)), 0: lambda: out.__setitem__(0, bases[0].multi_exp_prepared(S)),
                1: lambda: out.__setitem__(1, bases[1].multi_exp_prepared(S)), 2: lambda: out.__setitem__(2, bases[2].multi_exp_prepared(S, skip=npub))}
        th = [threading.Thread(target=jobs[k]) for k in (3, 4, 0, 1, 2)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        ms = (time.perf_counter() - t) * 1e3
        S.free()
        return out, ms

    ntt7(), msm5_seq(), msm5_conc(), ntt7_chains(), msm5_prepared(), prove_call_pattern()  # warm: domain tables, workspaces
    reps = 3
    chains_ms = float(np.mean([ntt7_chains() for _ in range(reps)]))
    prep = [msm5_prepared() for _ in range(reps)]
    prep_ms = float(np.mean([t for _, t in prep]))
    pat = [prove_call_pattern() for _ in range(reps)]
    pat_ms = float(np.mean([t for _, t in pat]))
    ntt_reps = [ntt7() for _ in range(reps)]
    if os.environ.get("ZKMI_BENCH_DEBUG"):
        print("inner boundary, zk_bn254_ntt per call and repetition (ms):", [[round(x, 2) for x in r] for r in ntt_reps], file=sys.stderr)
    ntt_ms = np.mean(ntt_reps, axis=0)
    seq = [msm5_seq() for _ in range(reps)]
    msm_ms = np.mean([t for _, t in seq], axis=0)
    conc = [msm5_conc() for _ in range(reps)]
    conc_ms = float(np.mean([t for _, t in conc]))
    # the proof from the five affine results: XYZZ records (x, y, 1, 1), infinity = all zero
    one = np.array([0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f], dtype=np.uint64)   # 1 in Montgomery form (Fp)
    zero = np.zeros(4, np.uint64)

    def record(res):
        parts = []
        for k, p in enumerate(res):
            if not p.any():
                parts.append(np.zeros(32 if k == 4 else 16, np.uint64))
            else:
                parts.append(np.concatenate([p, one, zero, one, zero] if k == 4 else [p, one, one]))
        return np.concatenate(parts)

    ok = all(par.groth16_finalize(inst.pk, record(res)[None, :], inst.r, inst.s) == proof for res in (seq[-1][0], conc[-1][0], prep[-1][0], pat[-1][0]))
    for bs in bases:
        bs.free()
    return {"what": "the same 2^%d proof's hot operations through the inner C ABI with host slices, call by call (zk_bn254_ntt x7, zk_bn254_msm_bases x5)" % inst.log_n,
            "ntt_calls_ms": [round(float(x), 3) for x in ntt_ms], "ntt_total_ms": round(float(ntt_ms.sum()), 3),
            "msm_calls_ms": dict(zip(("A", "B1", "K", "Z", "B2"), (round(float(x), 3) for x in msm_ms))), "msm_total_sequential_ms": round(float(msm_ms.sum()), 3),
            "msm_total_five_threads_ms": round(conc_ms, 3), "total_unpatched_call_sites_ms": round(float(ntt_ms.sum()) + conc_ms, 3),
            "ntt_total_three_chains_ms": round(chains_ms, 3), "msm_total_prepared_scalars_five_threads_ms": round(prep_ms, 3),
            "total_ms": round(chains_ms + prep_ms, 3), "total_ms_is": "INTEGRATION.md 4's patch: computeH's three chains on three goroutines + scalars registered once for A, B1, K, G2.B; the two groups one after the other",
            "overlapped_as_groth16_prove_issues_them_ms": round(pat_ms, 3),
            "overlapped_is": "the same calls with computeH on its own goroutine beside the wire-value MultiExps, Z after h (gnark v0.8.0 groth16 prove.go): wall time of the pattern",
            "reps": reps,
            "bytes_over_pcie_per_proof": int(7 * 2 * N * 32 + 5 * N * 32), "bases_register_s_once_per_key": round(reg_s, 3),
            "proof_from_these_results_matches_single_call": bool(ok),
            "note": "PCIe-inclusive (never `value`); excludes gnark's own Go code between the calls (the pointwise step of computeH, the solver, the host tail)"}
