#!/usr/bin/env python3
"""bench.py -- Groth16 prove (BN254) on synthetic R1CS data.

N = 1 (default): BASELINE.json configs[1], "Synthetic R1CS 2^20 constraints, BN254 Groth16 prove on 1xMI355X (G1 MSM + Fr NTT)".
One step = one proof from the solver output onwards (computeH = 7 NTTs, 4 G1 MSMs, 1 G2 MSM, host tail), inputs already resident
in HBM.  The same JSON line also carries, measured in the same run:
  * `prove_ms_host_inputs`  the proof through the entry point a cgo caller has (host slices: a, b, c, w cross PCIe inside the call),
  * `at_2p24`               the metric's other size (2^24 constraints on this one GPU: key with 84 GB of window tables), whose proof
                            bytes are checked against the recombination of two half-size slices run through the table-less path
                            (another window width, Horner, other task sizes) -- `--verify-2p24-oracle` adds the CPU oracle's bytes.
--gpus N > 1 (one process per GPU, torchrun): ONE proof over N * 2^21 constraints, range-sharded -- N = 8 is BASELINE.json
configs[2] (2^24 constraints).  Every rank owns one block of a, b, c, w, h and of the proving key (with its window tables); computeH
is block-sharded (the top log2 N butterfly stages of each transform run on all-to-all-transposed data: 10 RCCL all_to_all_single
per proof), the five MSMs run on the rank's slice, and an all-gather of the 768-byte partial-sum record lets every rank finish.

Prints ONE JSON line on rank 0 (contract in the task statement): metric / value / unit follow BASELINE.json; `roofline` describes the
dominant kernel (hipEvent pairs recorded inside libzkmi on the stream the kernels run on, live over the timed region);
`cpu_baseline` is the CPU oracle (a restatement of gnark's algorithm, NOT the gnark binary: no Go toolchain here) timed on this
box's host cores on the SAME 2^20 proof -- whose bytes are compared with the GPU's (parity at full size).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

GOLDEN = 0x9E3779B97F4A7C15
MASK = (1 << 64) - 1
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_PUBLIC = 8


def seed_at(seed: int, per: int, offset: int) -> int:
    """SplitMix64 stream `seed` advanced so that element 0 of the new stream is element `offset` of the old one
    (per = outputs consumed per element: 4 uniform, 5 witness-like)."""
    return (seed + per * offset * GOLDEN) & MASK


class Instance:
    """One synthetic proving instance: valid curve points P_i = k_i * G as the key, a, b uniform, c = a*b on the
    evaluation domain (h is a true quotient), w uniform or witness-like.  With world > 1 this is the rank's slice starting at `lo`."""

    def __init__(self, L, lib, zk, log_n, lo, n_public, witness, tables, shard_full_z=False, full_inputs=True, torch=None, window_shard=None,
                 abc_block=None):
        """window_shard=(rank, world): the whole key on every rank with this rank's table rows; abc_block=(lo, n): this rank's block of a, b, c."""
        self.L, self.lib, self.log_n, self.n_public = L, lib, log_n, n_public
        N = self.N = 1 << log_n
        dev = lib.DeviceBuffer

        def gen(fn, n, esz, seed, off):
            b = dev(n * esz)
            lib.check(fn(C.c_void_p(b.ptr), C.c_size_t(n), C.c_uint64(seed_at(seed, 4, off)), None))
            return b

        def gen_fr(seed, n, off, wit=0):
            b = dev(n * 32)
            lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(b.ptr), C.c_size_t(n), C.c_uint64(seed_at(seed, 5 if wit else 4, off)), C.c_int(1), C.c_int(wit), None))
            return b

        self.g1_a, self.g1_b, self.g1_k, self.g1_z = (gen(L.zk_bn254_g1_generate_dev, N, 64, s, lo) for s in (0xA1, 0xB1, 0xC1, 0xD1))
        self.g2_b = gen(L.zk_bn254_g2_generate_dev, N, 128, 0xB2, lo)
        self.small = {k: gen(L.zk_bn254_g1_generate_dev, 1, 64, s, 0).to_numpy(np.uint64, (8,)) for k, s in (("alpha", 1), ("beta", 2), ("delta", 3))}
        self.small2 = {k: gen(L.zk_bn254_g2_generate_dev, 1, 128, s, 0).to_numpy(np.uint64, (16,)) for k, s in (("beta", 8), ("delta", 9))}
        self.d_w = gen_fr(0xC, N, lo, witness)
        rs = gen_fr(0x23, 2, 0).to_numpy(np.uint64, (2, 4))  # pinned prover randomness (r, s)
        self.r, self.s = rs[0].copy(), rs[1].copy()
        self.d_a = self.d_b = self.d_c = None
        self.t_abc = None
        if full_inputs:
            self.d_a, self.d_b = gen_fr(0xA, N, lo), gen_fr(0xB, N, lo)
            self.d_c = dev(N * 32)
            lib.check(L.zk_bn254_fr_mul_dev(C.c_void_p(self.d_c.ptr), C.c_void_p(self.d_a.ptr), C.c_void_p(self.d_b.ptr), C.c_size_t(N), None))
        else:
            # this rank's blocks of a, b, c live in torch tensors (RCCL moves them)
            blo, bn = abc_block if abc_block else (lo, N)
            self.t_abc = [torch.empty((bn, 4), dtype=torch.int64, device="cuda") for _ in range(3)]
            for t, sd in zip(self.t_abc[:2], (0xA, 0xB)):
                lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(t.data_ptr()), C.c_size_t(bn), C.c_uint64(seed_at(sd, 4, blo)), C.c_int(1), C.c_int(0), None))
            lib.check(L.zk_bn254_fr_mul_dev(C.c_void_p(self.t_abc[2].data_ptr()), C.c_void_p(self.t_abc[0].data_ptr()), C.c_void_p(self.t_abc[1].data_ptr()),
                                            C.c_size_t(bn), None))
        self.pk = zk.ProvingKey(log_n, N, n_public, self.small["alpha"], self.small["beta"], self.small["delta"], self.g1_a, self.g1_b,
                                self.g1_k.ptr + n_public * 64, self.g1_z, self.small2["beta"], self.small2["delta"], self.g2_b,
                                bases_on_device=True, precompute_tables=tables, shard_full_z=shard_full_z, window_shard=window_shard,
                                table_window_bits=int(os.environ.get("ZKMI_BENCH_KEY_C", "0")))  # the variable: window-width sweeps (tooling)
        lib.check(L.zk_dev_sync())

    def g1_units(self):
        # G1 scalar-muls per proof: A (n), B1 (n), K (n - n_public), Z (N - 1); G2: B2 (n)
        return self.N + self.N + (self.N - self.n_public) + (self.N - 1)

    def free(self):
        self.pk.free()
        for b in (self.g1_a, self.g1_b, self.g1_k, self.g1_z, self.g2_b, self.d_w, self.d_a, self.d_b, self.d_c):
            if b is not None:
                b.free()


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


def oracle_proof(inst, log_n):
    """The CPU oracle proves the SAME instance on this box's host cores (test infrastructure: the checker and the timed CPU baseline)."""
    from oracle import oracle as orc  # the CPU oracle is used ONLY in these legs, after the timed GPU region
    N, npub = inst.N, inst.n_public
    cores = orc.max_threads()
    pkd = dict(log_domain=log_n, n_wires=N, n_public=npub, g1_alpha=inst.small["alpha"], g1_beta=inst.small["beta"], g1_delta=inst.small["delta"],
               g1_a=inst.g1_a.to_numpy(np.uint64, (N, 8)), g1_b=inst.g1_b.to_numpy(np.uint64, (N, 8)),
               g1_k=inst.g1_k.to_numpy(np.uint64, (N, 8))[npub:], g1_z=inst.g1_z.to_numpy(np.uint64, (N, 8)),
               g2_beta=inst.small2["beta"], g2_delta=inst.small2["delta"], g2_b=inst.g2_b.to_numpy(np.uint64, (N, 16)))
    ha, hb, hc, hw = (d.to_numpy(np.uint64, (N, 4)) for d in (inst.d_a, inst.d_b, inst.d_c, inst.d_w))
    t0 = time.perf_counter()
    cpu_proof, _ = orc.groth16_prove(pkd, ha, hb, hc, hw, inst.r, inst.s, nthreads=cores)
    return cpu_proof, time.perf_counter() - t0, cores


def plonk_block(L, lib, log_n, reps=int(os.environ.get("ZKMI_BENCH_PLONK_REPS", "3"))):  # the variable: A/B runs of tools/ab_bench.py that need a quieter figure
    """BASELINE.json configs[3]: "PLONK prove path (KZG-commit MSMs + coset NTTs) at 2^22 gates, 1xMI355X" -- the reference's only live
    prove path (plonk.Prove, backend/plonk/plonk.go:67).  Device-generated KZG SRS (real powers of alpha: kzg.NewSRS), a synthetic
    satisfiable circuit of 2^log_n rows (random wiring, random selectors, qK fixed per gate), plonk.Setup and plonk.Prove on the
    device; the 548 proof bytes are then handed to the CPU oracle's VERIFIER (quotient identity + two KZG pairing checks)."""
    from noir_backend_using_gnark_amd import bn254 as zb, plonk as zp
    n = 1 << log_n
    npub, nvars = 4, n // 2
    nc = n - npub
    alpha = 0xA1FA0123456789ABCDEF
    t0 = time.time()
    d_srs = lib.DeviceBuffer((n + 3) * 64)
    a_m = np.frombuffer((alpha * (1 << 256) % R_FR).to_bytes(32, "little"), dtype=np.uint64).copy()
    lib.check(L.zk_bn254_kzg_new_srs_dev(C.c_void_p(d_srs.ptr), C.c_size_t(n + 3), lib.vp(a_m), None, None))
    srs = zb.ResidentBases(d_srs, n=n + 3, table_window_bits=int(os.environ.get("ZKMI_BENCH_SRS_C", "0")))  # the variable: window-width sweeps (tooling)
    rng = np.random.default_rng(5)
    xa, xb, xc = (rng.integers(0, nvars, nc, dtype=np.uint32) for _ in range(3))
    dsol = lib.DeviceBuffer(nvars * 32)
    lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(dsol.ptr), C.c_size_t(nvars), C.c_uint64(0x51), C.c_int(1), C.c_int(1), None))
    coef = []
    for sd in (1, 2, 3, 4):
        b = lib.DeviceBuffer(nc * 32)
        lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(b.ptr), C.c_size_t(nc), C.c_uint64(sd), C.c_int(1), C.c_int(0), None))
        coef.append(b)
    dqk = lib.DeviceBuffer(nc * 32)
    dx = [lib.DeviceBuffer.from_numpy(v) for v in (xa, xb, xc)]
    lib.check(L.zk_bn254_plonk_synth_qk_dev(C.c_void_p(dqk.ptr), *[C.c_void_p(b.ptr) for b in coef], *[C.c_void_p(b.ptr) for b in dx], C.c_void_p(dsol.ptr),
                                            C.c_size_t(nc), None))
    t_data = time.time() - t0
    t0 = time.time()
    pk = zp.setup(zp.Circuit(npub, nvars, coef[0], coef[1], coef[2], coef[3], dqk, xa, xb, xc), srs)
    t_setup = time.time() - t0
    bl = np.arange(1, 37, dtype=np.uint64).reshape(9, 4)  # any nine scalars < r (Montgomery images of something)
    proof = zp.prove(pk, dsol, bl)
    # One-time precomputation per key, like the window tables: the SRS in Lagrange form over the key's domain (csrc/lagrange.hip), after which l, r, o are
    # committed from the wire values -- the same digests (the proof bytes are compared below).  The figure without it is kept beside the headline figure.
    lib.profile(True)  # the same conditions as the timed loop below (its event pairs cost the small sizes up to a millisecond)
    t0 = time.perf_counter()
    for _ in range(reps):
        proof_monomial = zp.prove(pk, dsol, bl)
    ms_monomial = (time.perf_counter() - t0) * 1e3 / reps
    lib.profile(False)
    lagrange_ms = None
    if os.environ.get("ZKMI_BENCH_PLONK_LAGRANGE", "1") != "0":  # the variable: A/B runs (tooling)
        t0 = time.perf_counter()
        pk.lagrange_srs()
        lagrange_ms = (time.perf_counter() - t0) * 1e3
        proof = zp.prove(pk, dsol, bl)
    lib.profile(True)
    lib.profile_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        proof = zp.prove(pk, dsol, bl)
    ms = (time.perf_counter() - t0) / reps * 1e3
    lib.profile(False)
    prof, host_sections = lib.split_profile(lib.profile_read())
    out = {"gates": n, "prove_ms": round(ms, 2), "steps": reps, "warmup": 1, "setup_ms": round(t_setup * 1e3, 1), "data_s": round(t_data, 2),
           "lro_commitments": "from the wire values against the SRS's Lagrange form (zk_bn254_plonk_pk_lagrange_srs, once per key)" if lagrange_ms is not None else "from coefficients",
           "lagrange_srs_ms_once_per_key": None if lagrange_ms is None else round(lagrange_ms, 1), "prove_ms_lro_from_coefficients": round(ms_monomial, 2),
           "same_bytes_both_ways": bool(proof == proof_monomial),
           # wall clock of the protocol's rounds (each ends in a digest the next challenge needs): where a proof's time goes when its kernels do not fill it
           "rounds_ms": {k.split(".", 1)[1]: round(v[1] / reps, 3) for k, v in host_sections.items() if k.startswith("plonk.")},
           "kzg_commits_per_proof": "10 (9 as MSMs; the linearised polynomial's digest by linearity from the verifying key and [Z])", "ntt_per_proof": "4 x inverse(n) + 4 x coset(4n) + 1 x coset inverse(4n) (gnark's fifth pair -- qk with the public inputs -- is one element-wise kernel here)",
           "proof_sha": hashlib.sha256(proof).hexdigest()[:16],
           "kernel_ms_per_proof": {k: round(v[1] / reps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:14]},
           # nine commitments of ~n scalars per proof against the SRS's window table; l, r, o and h1, h2, h3 are ONE accumulate launch each (three bucket sets),
           # so a launch carries 9n / (launches per proof) scalars on average; the transforms' passes work on n (small domain) or 4n points
           "roofline": block_roofline(prof, reps, 9.0 * n * reps / max(1, prof.get("msm_accumulate_g1", (reps * 9, 0.0))[0]), 0, 4 * n, log_n)}
    if out["roofline"]:
        out["roofline"]["scalar_muls_per_proof"] = 9 * n
    # ---- checker (CPU oracle, after the timed region): decode Proof.WriteTo and run plonk.Verify's equations
    from oracle import bn254_ref as ref, plonk_ref as pl

    def dec(b):
        if b[0] >> 6 == 1:
            return None
        x = int.from_bytes(bytes([b[0] & 0x3F]) + b[1:], "big")
        y = pow((x * x * x + 3) % ref.Q, (ref.Q + 1) // 4, ref.Q)
        return (x, ref.Q - y if (y > (ref.Q - 1) // 2) != (b[0] >> 6 == 3) else y)

    pts = [dec(proof[32 * i:32 * i + 32]) for i in range(7)]
    claimed = [int.from_bytes(proof[260 + 32 * i:292 + 32 * i], "big") for i in range(7)]
    pr = dict(lro=pts[0:3], z=pts[3], h=pts[4:7], batch_h=dec(proof[224:256]), claimed=claimed, z_open_h=dec(proof[484:516]), zu=int.from_bytes(proof[516:548], "big"))
    P = pl.g1_from_np
    vkd = pk.vk
    vk = dict(size=n, size_inv=ref.inv(n, ref.R), generator=pl.mont_np_to_ints(vkd["generator"])[0], n_public=npub, coset_shift=5,
              srs_g2=[ref.G2_GEN, ref.g2_mul(ref.G2_GEN, alpha)], s=[P(p) for p in vkd["s"]], ql=P(vkd["ql"]), qr=P(vkd["qr"]), qm=P(vkd["qm"]), qo=P(vkd["qo"]), qk=P(vkd["qk"]))
    pub = pl.mont_np_to_ints(dsol.to_numpy(np.uint64, (npub, 4)))
    out["verified_by"] = "oracle/plonk_ref.plonk_verify: Fiat-Shamir re-derived from the bytes, quotient identity at zeta, two KZG openings by pairings"
    out["proof_verifies"] = bool(pl.plonk_verify(vk, pr, pub))
    out["wrong_public_input_rejected"] = bool(not pl.plonk_verify(vk, pr, [(pub[0] + 1) % ref.R] + pub[1:]))
    # the product's own host-side verifier (zk_bn254_plonk_verify) on the same bytes, with the key image it would get from the wire: same verdicts, timed
    from noir_backend_using_gnark_amd import verify as zv
    g2 = np.stack([np.frombuffer(ref.g2_affine_mont_bytes(q), dtype=np.uint64) for q in vk["srs_g2"]])
    vkb = pl.plonk_vk_bytes(vk)
    pub_m = dsol.to_numpy(np.uint64, (npub, 4))
    t0 = time.perf_counter()
    acc = zv.plonk_verify(proof, vkb, g2, pub_m)
    out["host_verify"] = {"accepts": bool(acc), "ms": round((time.perf_counter() - t0) * 1e3, 2),
                          "rejects_wrong_public_input": bool(not zv.plonk_verify(proof, vkb, g2, pl.ints_to_mont_np([(pub[0] + 1) % ref.R] + pub[1:])))}
    pk.free()
    srs.free()
    return out


R_FR = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
HAND_WRITTEN_HOT = ("msm_accumulate_g1", "msm_accumulate_g2", "ntt_pass_contig", "ntt_pass_strided", "msm_sort_pass", "msm_sort_hist")


def dominant_by_time(prof, steps):
    """The kernel with the largest total time in the timed region, whatever it is -- library kernels included (rocPRIM's radix sort is timed as one event pair
    around the whole library call; every other entry is one kernel).  `roofline` below prices the dominant HAND-WRITTEN hot kernel; when this entry names
    another kernel, that one is the larger consumer of kernel time."""
    if not prof:
        return None
    name, (launches, tot) = max(prof.items(), key=lambda kv: kv[1][1])
    return {"kernel": name, "ms_per_step": round(tot / steps, 4), "launches_per_step": round(launches / steps, 2), "hand_written": "rocprim" not in name}


def block_roofline(prof, steps, g1_units_per_launch, g2_units_per_launch, ntt_elems_per_launch, log_key, tables=True):
    """`roofline` for one measured block (same definition as the headline's): dominant hand-written hot kernel by total time, achieved = algorithmic bytes
    per launch / average launch duration (event pairs inside libzkmi on the stream of the launch), traffic from profiles/pmc_traffic.json at that size."""
    hot = {k: v for k, v in prof.items() if k in HAND_WRITTEN_HOT[:4]}
    if not hot:
        return None
    name, (launches, tot_ms) = max(hot.items(), key=lambda kv: kv[1][1])
    per = tot_ms / launches
    units, bpu = {"msm_accumulate_g1": (g1_units_per_launch, 96.0), "msm_accumulate_g2": (g2_units_per_launch, 160.0)}.get(name, (ntt_elems_per_launch, 64.0))
    achieved = units * bpu / (per * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath) and tables:
        try:
            traffic = json.load(open(tpath)).get(name, {}).get("by_log_n", {}).get(str(log_key), {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    return {"kernel": name, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic, "avg_launch_ms": round(per, 4), "launches": launches, "units_per_launch": int(units), "algorithmic_bytes_per_unit": bpu,
            "dominant_by_time": dominant_by_time(prof, steps),
            "kernel_ms_per_step": {k: round(v[1] / steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:10]}}


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




def micro_block(L, lib, zk, log_n):
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
    dom = zk.Domain(n)
    head = sc.to_numpy(np.uint64, (4096, 4))
    dom.fft(sc, zk.DIF)
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


def single_process_main(args, zk, lib):
    """bench.py --gpus N --single-process: the multi-GPU path a caller of the C ABI gets WITHOUT becoming one process per GPU (the reference is one process:
    gnark_backend_ffi/main.go:24-37) -- zk_init_devices + a proving key with device_mask, then the ordinary zk_bn254_groth16_prove."""
    L = lib.lib()
    N = args.gpus
    real = max(1, int(L.zk_device_count()))
    devs = [i % real for i in range(N)]
    lib.check(L.zk_init_devices((C.c_int * N)(*devs), C.c_size_t(N)))
    lib.check(L.zk_set_default_devices(C.c_uint32(0)))
    log_n = args.log_n if args.log_n is not None else 20
    witness = 1 if args.scalars == "witness" else 0
    t0 = time.time()
    inst = Instance(L, lib, zk, log_n, 0, N_PUBLIC, witness, not args.no_tables)
    single = zk.prove(inst.pk, inst.d_a, inst.d_b, inst.d_c, inst.d_w, inst.r, inst.s, n_constraints=inst.N, on_device=True)
    inst.pk.free()
    pk = zk.ProvingKey(log_n, inst.N, N_PUBLIC, inst.small["alpha"], inst.small["beta"], inst.small["delta"], inst.g1_a, inst.g1_b, inst.g1_k.ptr + N_PUBLIC * 64, inst.g1_z,
                       inst.small2["beta"], inst.small2["delta"], inst.g2_b, bases_on_device=True, precompute_tables=not args.no_tables, device_mask=(1 << N) - 1)
    setup_s = time.time() - t0
    step = lambda: zk.prove(pk, inst.d_a, inst.d_b, inst.d_c, inst.d_w, inst.r, inst.s, n_constraints=inst.N, on_device=True)
    proof = None
    for _ in range(args.warmup):
        proof = step()
    lib.check(L.zk_dev_sync())
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proof = step()
    lib.check(L.zk_dev_sync())
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    metric = "Groth16 prove ms + BN254 G1 MSM scalar-muls/sec at 2^20 / 2^24 constraints"
    try:
        metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        pass
    out = {"metric": metric, "value": round(inst.g1_units() / (ms * 1e-3), 1), "unit": "G1 scalar-muls/s (whole prove: 4 G1 MSMs + G2 MSM + 7 NTTs per step)",
           "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "prove_ms": round(ms, 3), "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "u32", "data": "synthetic",
           "config": {"workload": "groth16_prove_bn254_synthetic_r1cs_2^%d" % log_n, "constraints": inst.N, "wires": inst.N, "n_public": N_PUBLIC, "scalars": args.scalars,
                      "parallelism": "ONE process, %d device entries behind zk_bn254_groth16_prove (composite key by wire range, block-sharded computeH, peer-copy "
                                     "transposes, 768-byte records combined on the host)" % N,
                      "devices": devs, "real_gpus": real, "virtual_entries": real < N},
           "proof_equals_single_entry": bool(proof == single), "proof_sha": hashlib.sha256(proof).hexdigest()[:16], "setup_s": round(setup_s, 2),
           "roofline": None, "cpu_baseline": None,
           "note": "virtual entries share one GPU: this line shows that the path runs and gives the single-GPU bytes, not how it scales" if real < N else None}
    if proof != single:
        out["parity_error"] = "the proof over %d device entries differs from the single-entry proof" % N
    pk.free()
    print(json.dumps(out))


def go_toolchain_probe():
    """is there a Go toolchain (and gnark's module cache) on this box?  If so tools/go_pin checks the committed fixtures against
    the real gnark / gnark-crypto (go.mod:5,23) and the counts are reported; otherwise the oracle stays the checker ("parity unpinned")."""
    import shutil
    import subprocess
    go = shutil.which("go")
    out = {"go": go, "version": None, "module_cache": None, "go_pin": None}
    if not go:
        return out
    try:
        out["version"] = subprocess.run([go, "version"], capture_output=True, text=True, timeout=30).stdout.strip()
        cache = subprocess.run([go, "env", "GOMODCACHE"], capture_output=True, text=True, timeout=30).stdout.strip()
        have = os.path.isdir(os.path.join(cache, "github.com", "consensys")) if cache else False
        out["module_cache"] = {"path": cache, "has_consensys_modules": have}
        if have:
            r = subprocess.run([go, "run", "."], cwd=os.path.join(ROOT, "tools", "go_pin"), capture_output=True, text=True, timeout=900,
                               env=dict(os.environ, GOFLAGS="-mod=mod", GOPROXY="off"))
            txt = r.stdout + r.stderr
            out["go_pin"] = {"rc": r.returncode, "pass": txt.count("PASS"), "fail": txt.count("FAIL"), "tail": txt[-400:]}
    except Exception as e:
        out["error"] = str(e)[:200]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed proofs (default 200: ~2 s of GPU work at 2^20, enough for a utilisation sampler to see)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log-n", type=int, default=None, help="log2(constraints) per GPU (default: 20 on one GPU = configs[1]; 21 per GPU on several, so that 8 GPUs prove configs[2]'s 2^24)")
    ap.add_argument("--scalars", choices=["uniform", "witness"], default="uniform")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tables", action="store_true", help="disable the precomputed window tables of the resident proving key")
    ap.add_argument("--shard", choices=["range", "windows"], default="range", help="multi-GPU decomposition of the MSMs: by point range (default: no scalar "
                    "exchange) or by digit window / table row (north_star wording: whole key on every rank, h all-gathered)")
    ap.add_argument("--force-sharded", action="store_true", help="run the multi-GPU decomposition (sharded computeH phases + msm5_pk + all-gather + finalize) even at N=1")
    ap.add_argument("--no-2p24", action="store_true", help="skip the second measured block (2^24 constraints on this GPU)")
    ap.add_argument("--no-host-inputs", action="store_true", help="skip the host-input (PCIe-inclusive) measurement")
    ap.add_argument("--no-plonk", action="store_true", help="skip the PLONK block (configs[3]: plonk.Prove at 2^22 gates, verified by the oracle's pairing verifier)")
    ap.add_argument("--plonk-log-n", type=int, default=22)
    ap.add_argument("--no-micro", action="store_true", help="skip the configs[4] block (2^26-point G1 MSM + 2^26 NTT) and the SRS-load block")
    ap.add_argument("--micro-log-n", type=int, default=26, help="log2 of the points of the sharded configs[4] block at N > 1 (total over all ranks)")
    ap.add_argument("--single-process", action="store_true", help="with --gpus N: ONE process drives N device entries through the same zk_bn254_groth16_prove call (csrc/multidev.hip: "
                    "composite key by wire range, block-sharded computeH, peer-copy transposes); no torch.distributed.  On a box with fewer GPUs the devices are listed "
                    "repeatedly (virtual entries: correctness of the path, not a scaling figure).  --log-n is the TOTAL size")
    ap.add_argument("--lib", default=None, help="measurement tooling: 'exp' binds this run to libzkmi_exp.so (the A/B switches of DESIGN.md 8), or a path to another build")
    ap.add_argument("--no-export", action="store_true", help="skip the export-path block (PlonkPreprocess -> PlonkProveWithPK -> PlonkVerifyWithVK through libgnark_backend.so at 2^19 gates)")
    ap.add_argument("--export-log-gates", type=int, default=19)
    ap.add_argument("--verify-2p24-oracle", action="store_true", help="also check the 2^24 proof bytes against the CPU oracle (~2 min on 128 cores)")
    args = ap.parse_args()

    import torch
    import noir_backend_using_gnark_amd as zk
    from noir_backend_using_gnark_amd import _lib, parallel as par
    if args.lib:
        _lib.use_library(os.path.join(ROOT, "noir_backend_using_gnark_amd", "csrc", "build_exp", "libzkmi_exp.so") if args.lib == "exp" else args.lib)

    if args.single_process:
        return single_process_main(args, zk, _lib)
    rank, world, local = par.init_distributed()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    L = _lib.lib()
    local = local % max(1, torch.cuda.device_count())  # ranks share a GPU only in the gloo dry run on a one-GPU box
    _lib.check(L.zk_init(C.c_int(local)))
    torch.cuda.set_device(local)
    _lib.require_device()

    log_n = args.log_n if args.log_n is not None else (20 if world == 1 else 21)
    log_ng = log_n + (world.bit_length() - 1)
    if (1 << (log_ng - log_n)) != world:
        raise SystemExit("--gpus must be a power of two")
    N_loc, N_g = 1 << log_n, 1 << log_ng
    witness = 1 if args.scalars == "witness" else 0
    lo, hi = par.shard_range(N_g, rank, world)  # this rank's slice of every wire-indexed / coefficient-indexed array
    assert hi - lo == N_loc
    sharded = world > 1 or args.force_sharded
    np_loc = N_PUBLIC if rank == 0 else 0  # public wires live in rank 0's slice; gnark's pk.G1.K starts at the first private wire

    t_setup = time.time()
    win = sharded and args.shard == "windows"
    if win:  # every rank: the whole key (its own table rows), the whole w, its block of a, b, c
        inst = Instance(L, _lib, zk, log_ng, 0, N_PUBLIC, witness, True, full_inputs=False, torch=torch, window_shard=(rank, world), abc_block=(lo, N_loc))
    else:
        inst = Instance(L, _lib, zk, log_n, lo, np_loc if sharded else N_PUBLIC, witness, not args.no_tables,
                        shard_full_z=(sharded and rank != world - 1), full_inputs=not sharded, torch=torch)
    pk, d_w, r, s = inst.pk, inst.d_w, inst.r, inst.s
    fallback_stream = torch.cuda.Stream(priority=-1) if sharded else None
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    def step():
        if not sharded:
            return zk.prove(pk, inst.d_a, inst.d_b, inst.d_c, d_w, r, s, n_constraints=N_g, on_device=True)
        sess = par.groth16_msm5_pk_begin(pk, d_w.ptr)  # digits / sort / task plan of w run under computeH and its exchanges
        try:
            try:
                side = torch.cuda.ExternalStream(par.groth16_session_stream(sess))  # the library's computeH stream, shared with torch / RCCL
            except Exception:  # a torch build without ExternalStream: any non-null stream works (the library then bridges with events)
                side = fallback_stream
            with torch.cuda.stream(side):
                # the prover consumes its working buffers: at N > 1 the first all-to-all already writes fresh ones, at N = 1 copy
                a, b, c = (t.clone() for t in inst.t_abc) if world == 1 else inst.t_abc
                h = par.compute_h_sharded(a, b, c, log_ng, rank, world)
                if win:
                    h = par.all_gather_blocks(h)  # the coefficient exchange of the window mode: every rank needs the whole h
                live, sess = sess, None  # _end releases the session whatever it returns
                rec = par.groth16_msm5_pk_end(live, h.data_ptr(), side.cuda_stream)
                return par.groth16_finalize(pk, par.all_gather_limbs(rec), r, s)
        finally:
            if sess is not None:  # computeH or an exchange raised between _begin and _end: give the five stream slots back
                par.groth16_msm5_pk_abort(sess)

    def barrier():
        if world > 1:
            par.dist().barrier()
        torch.cuda.synchronize()
        _lib.check(L.zk_dev_sync())

    proof = None
    for _ in range(args.warmup):
        proof = step()
    _lib.profile(True)
    _lib.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proof = step()
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.profile(False)
    prof = _lib.profile_read()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if par.dist().get_backend() == "nccl" else "cpu")
        par.dist().all_reduce(t, op=par.dist().ReduceOp.MAX)
        elapsed = float(t.item())
        allp = par.all_gather_limbs(np.frombuffer(proof, dtype=np.uint64))
        assert (allp == allp[0]).all(), "ranks disagree on the proof bytes"

    g1_units = N_g + N_g + (N_g - N_PUBLIC) + (N_g - 1)
    ms_per_step = elapsed / args.steps * 1e3
    value = g1_units * args.steps / elapsed

    # ---- roofline of the dominant kernel (by total time over the timed region)
    # restricted to the hand-written hot kernels of the path; with the five MSMs of a proof running on concurrent
    # streams an event pair also sees the time a kernel spends sharing the machine, exactly as rocprofv3 does
    hot = {k: v for k, v in prof.items() if k in HAND_WRITTEN_HOT[:4]}
    name, (launches, tot_ms) = max(hot.items(), key=lambda kv: kv[1][1])
    per_launch_ms = tot_ms / launches
    if name == "msm_accumulate_g1":
        # the G1 MSMs of a proof come in launches / steps accumulate launches (A, B1, K share one: same sorted digits)
        units_per_launch, bytes_per_unit = g1_units / world / (launches / float(args.steps)), 96.0
    elif name == "msm_accumulate_g2":
        units_per_launch, bytes_per_unit = N_loc, 160.0
    else:  # an NTT pass: 64 B per element per pass
        units_per_launch, bytes_per_unit = (N_loc if sharded else N_g), 64.0
    achieved = units_per_launch * bytes_per_unit / (per_launch_ms * 1e-3) / 1e9
    # HBM traffic per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 correction:
    # tools/pmc_traffic.py) -- taken at the SAME per-GPU size and table setting, else null
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath) and not args.no_tables:
        try:
            traffic = json.load(open(tpath)).get(name, {}).get("by_log_n", {}).get(str(log_n), {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    # the honest ceiling of this kernel is the VALU, not HBM: mixed additions per second against the measured peak of the
    # mixed-addition routine alone (tools/ubench.hip: k_madd29 17.4 G/s, G1; a G2 mixed addition costs ~2.1 G1 ones)
    valu = None
    if name.startswith("msm_accumulate"):
        wb, dg = C.c_uint32(0), C.c_uint32(0)
        _lib.check(L.zk_bn254_msm_plan_info(C.c_size_t(N_g if win else N_loc), C.c_int(1 if pk.info()["tables"] else 0), C.byref(wb), C.byref(dg)))
        digits = int(dg.value)
        madds = units_per_launch * digits
        peak = 17.4e9 if name.endswith("g1") else 17.4e9 / 2.1
        valu = {"mixed_adds_per_launch": int(madds), "achieved_madd_per_s": round(madds / (per_launch_ms * 1e-3), 1), "peak_madd_per_s": peak,
                "frac": round(madds / (per_launch_ms * 1e-3) / peak, 4), "window_bits": int(wb.value), "peak_source": "tools/ubench.hip k_madd29, 4 waves/SIMD: 17.44 G madd/s (9,017 cycles per wave)"}
    roofline = {"kernel": name, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "avg_launch_ms": round(per_launch_ms, 4), "launches": launches, "valu": valu, "dominant_by_time": dominant_by_time(prof, args.steps),
                "note": "VALU-bound kernel (254-bit modular multiplies on 32-bit integer ALUs); see DESIGN.md for the ALU-issue fraction",
                "kernel_ms_per_step": {k: round(v[1] / args.steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}}

    # the metric's name is BASELINE.json's, verbatim; the size this run used is in config.workload / config.constraints
    baseline_metric = "Groth16 prove ms + BN254 G1 MSM scalar-muls/sec at 2^20 / 2^24 constraints"
    try:
        baseline_metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        pass
    if log_ng == 20 and world == 1:
        bcfg = "BASELINE.json configs[1]: Synthetic R1CS 2^20 constraints, BN254 Groth16 prove on 1xMI355X (G1 MSM + Fr NTT)"
    elif log_ng == 24 and world == 8:
        bcfg = "BASELINE.json configs[2]: Synthetic R1CS 2^24 constraints, G1+G2 MSM sharded across 8xMI355X via RCCL/xGMI (2^21 per GPU)"
    elif world > 1:
        bcfg = "BASELINE.json configs[2] shape at %d GPUs: 2^%d constraints, 2^%d per GPU (weak scaling towards 8 x 2^21 = 2^24)" % (world, log_ng, log_n)
    else:
        bcfg = "configs[1] workload at 2^%d" % log_ng
    out = {
        "metric": baseline_metric,
        "value": round(value, 1), "unit": "G1 scalar-muls/s (whole prove: 4 G1 MSMs + G2 MSM + 7 NTTs per step)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "prove_ms": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": "groth16_prove_bn254_synthetic_r1cs_2^%d" % log_ng, "baseline_config": bcfg, "constraints": N_g, "wires": N_g,
                   "n_public": N_PUBLIC, "scalars": args.scalars, "per_gpu_constraints": N_loc, "window_tables": bool(pk.info()["tables"]),
                   "parallelism": "single GPU" if not sharded else
                   ("one proof window-sharded x%d: block-sharded computeH (all-to-all transposes), all-gather of h, MSMs over all wires on this rank's table rows "
                    "(all-gather of partial sums)" % world) if win else
                   "one proof range-sharded x%d: block-sharded computeH (all-to-all transposes) + MSMs on rank-local key slices (all-gather of partial sums)" % world},
        "roofline": roofline, "proof_sha": hashlib.sha256(proof).hexdigest()[:16], "setup_s": round(t_setup, 2),
    }
    if par.dist().is_available() and par.dist().is_initialized():
        out["config"]["collectives"] = par.dist().get_backend()  # "nccl" = RCCL carried the exchanges of this run

    single = rank == 0 and not sharded
    # ---- the same proof through the entry point a cgo caller has: host slices in, 128 bytes out (PCIe-inclusive; never `value`)
    if single and not args.no_host_inputs:
        ha, hb, hc, hw = (d.to_numpy(np.uint64, (N_g, 4)) for d in (inst.d_a, inst.d_b, inst.d_c, inst.d_w))
        zk.prove(pk, ha, hb, hc, hw, r, s)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            hp = zk.prove(pk, ha, hb, hc, hw, r, s)
        host_ms = (time.perf_counter() - t0) / reps * 1e3
        out["prove_ms_host_inputs"] = {"value": round(host_ms, 3), "reps": reps, "bytes_over_pcie": int(4 * N_g * 32), "proof_matches_device_inputs": bool(hp == proof),
                                       "note": "zk_bn254_groth16_prove(on_device=0): a, b, c, w are pageable host arrays uploaded inside the call"}
        del ha, hb, hc, hw

    # ---- the same proof's hot operations through the INNER boundary, call by call, with host slices (north_star's literal flow)
    if single and not args.no_host_inputs and log_n <= 22:
        out["inner_boundary_2p%d" % log_n] = inner_boundary_block(L, _lib, zk, par, inst, proof)
        if not out["inner_boundary_2p%d" % log_n]["proof_from_these_results_matches_single_call"]:
            out["parity_error"] = "inner boundary: the proof assembled from separate MultiExp calls differs from the single-call prover's"

    # ---- the same key with a witness-like wire vector (SURVEY 8d cfg2's second distribution: 50 % in {0, 1}, 25 % < 2^32, 25 % uniform): giant buckets, split tasks, folds
    if single and not args.no_host_inputs and args.scalars == "uniform":
        d_w2 = _lib.DeviceBuffer(N_g * 32)
        _lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(d_w2.ptr), C.c_size_t(N_g), C.c_uint64(seed_at(0xC, 5, 0)), C.c_int(1), C.c_int(1), None))
        run_w = lambda: zk.prove(pk, inst.d_a, inst.d_b, inst.d_c, d_w2, r, s, n_constraints=N_g, on_device=True)
        pw0 = run_w()
        reps = 20
        _lib.check(L.zk_dev_sync())
        t0 = time.perf_counter()
        for _ in range(reps):
            pw = run_w()
        w_ms = (time.perf_counter() - t0) / reps * 1e3
        out["prove_ms_witness_like_scalars"] = {"value": round(w_ms, 3), "reps": reps, "deterministic": bool(pw == pw0),
                                                "note": "same key and a, b, c; wire values 50 % in {0, 1}, 25 % below 2^32, 25 % uniform (bytes checked against the oracle in tests/test_gpu_parity.py at 2^14)"}
        d_w2.free()

    # ---- CPU baseline: the oracle proves the SAME instance on this box's host cores (rank 0, N=1 only)
    if single and not args.no_cpu_baseline:
        cpu_proof, cpu_s, cores = oracle_proof(inst, log_n)
        out["cpu_baseline"] = {"value": round(g1_units / cpu_s, 1), "unit": out["unit"], "cores": cores, "kind": "port",
                               "sample": "1 full proof of the same instance (2^%d constraints, same seeds) by oracle/bn254_oracle.c "
                                         "(OpenMP, window-parallel Pippenger + radix-2 FFT; a restatement, not the gnark binary)" % log_n,
                               "prove_ms": round(cpu_s * 1e3, 1), "proof_bytes_match_gpu": bool(cpu_proof == proof)}
        if cpu_proof != proof:
            out["parity_error"] = "GPU proof bytes differ from the CPU oracle's"

    # ---- second measured block: the metric's other size, 2^24 constraints on this GPU (BASELINE metric "at 2^20 / 2^24 constraints")
    if single and log_n == 20 and not args.no_2p24 and not args.no_tables:
        inst.free()
        t1 = time.time()
        big = Instance(L, _lib, zk, 24, 0, N_PUBLIC, witness, True)
        setup24 = time.time() - t1
        run24 = lambda: zk.prove(big.pk, big.d_a, big.d_b, big.d_c, big.d_w, big.r, big.s, n_constraints=big.N, on_device=True)
        p24 = run24()
        _lib.check(L.zk_dev_sync())
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            p24 = run24()
        _lib.check(L.zk_dev_sync())
        ms24 = (time.perf_counter() - t0) / reps * 1e3  # timed WITHOUT the per-launch event pairs of the profile ...
        _lib.profile(True)
        _lib.profile_reset()
        for _ in range(reps):                             # ... which come from the same number of proofs run once more
            run24()
        _lib.check(L.zk_dev_sync())
        _lib.profile(False)
        prof24 = _lib.profile_read()
        blk = {"constraints": big.N, "prove_ms": round(ms24, 2), "value": round(big.g1_units() / (ms24 * 1e-3), 1), "unit": out["unit"], "steps": reps,
               "warmup": 1, "setup_s": round(setup24, 2), "proof_sha": hashlib.sha256(p24).hexdigest()[:16], "window_tables": bool(big.pk.info()["tables"]),
               "verified_by": "two-slice recombination through the table-less msm5 path + finalize (independent window width / Horner / task sizes)",
               "proof_bytes_match_recombination": bool(two_slice_recombination(big, par, _lib, L) == p24)}
        n24 = prof24.get("msm_accumulate_g1", (1, 0))[0]
        blk["roofline"] = block_roofline(prof24, reps, big.g1_units() / max(1.0, n24 / float(reps)), big.N, big.N, 24)
        if not blk["proof_bytes_match_recombination"]:
            out["parity_error"] = "2^24: single-call proof differs from the two-slice recombination"
        if args.verify_2p24_oracle:
            cpu_proof, cpu_s, cores = oracle_proof(big, 24)
            blk["cpu_oracle"] = {"prove_ms": round(cpu_s * 1e3, 1), "cores": cores, "proof_bytes_match_gpu": bool(cpu_proof == p24)}
            if cpu_proof != p24:
                out["parity_error"] = "2^24: GPU proof bytes differ from the CPU oracle's"
        out["at_2p24"] = blk
        big.free()
        inst = None
    # ---- third measured block: BASELINE configs[3], the PLONK prover at 2^22 gates
    if single and log_n == 20 and not args.no_plonk:
        if inst is not None:
            inst.free()
        out["plonk_2p%d" % args.plonk_log_n] = plonk_block(L, _lib, args.plonk_log_n)
        if not out["plonk_2p%d" % args.plonk_log_n]["proof_verifies"]:
            out["parity_error"] = "PLONK: the proof does not verify"
        inst = None
    # ---- configs[4] (standalone 2^26 MSM / NTT) and the SRS load of the reference's size
    if single and log_n == 20 and not args.no_micro:
        if inst is not None:
            inst.free()
            inst = None
        out["micro_2p26"] = micro_block(L, _lib, zk, 26)
        out["srs_read_1e6"] = srs_block(_lib)
        if not (out["micro_2p26"]["equals_split_recombination"] and out["micro_2p26"]["equals_window_table_path"] and out["srs_read_1e6"]["write_of_read_is_identity"]):
            out["parity_error"] = "micro-benchmark cross-check failed"
    # ---- the reference's live call end to end through libgnark_backend.so (child processes; the GPU is shared with this one, which is idle meanwhile)
    if single and log_n == 20 and not args.no_export:
        if inst is not None:
            inst.free()
            inst = None
        try:
            out["export_path"] = export_path_block(args.export_log_gates)
            if not out["export_path"]["ok"]:
                out["parity_error"] = "export path: a proof made through libgnark_backend.so does not verify"
        except Exception as e:
            out["export_path"] = {"error": str(e)[:600]}
    if rank == 0:
        out["go_toolchain"] = go_toolchain_probe()
    # ---- configs[4] on several GPUs: the 2^26-point MSM range-sharded and the 2^26-point FFT block-sharded over the ranks
    if (world > 1 or (args.force_sharded and par._force_collectives())) and not args.no_micro:
        inst.free()
        blk = micro_sharded_block(L, _lib, zk, par, torch, args.micro_log_n, rank, world)
        out["micro_2p%d_sharded" % args.micro_log_n] = blk
        if not (blk["msm_same_on_every_rank"] and blk["msm_equals_odd_split_recombination"] and blk["ntt_inverse_of_forward_is_identity"]):
            out["parity_error"] = "sharded micro-benchmark cross-check failed"
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        par.dist().barrier()
        par.dist().destroy_process_group()


if __name__ == "__main__":
    main()
