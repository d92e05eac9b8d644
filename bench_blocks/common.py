"""bench.py's shared pieces: the synthetic instance of BASELINE.json configs[1] / [2] (random valid proving key and solver output built on the device), the CPU
oracle's proof of the same instance (test infrastructure: only called AFTER the timed region), and the roofline arithmetic of the bench line."""
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

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


def valu_fraction(L, lib, name, units_per_launch, per_launch_ms, n_plan, tables):
    """The honest ceiling of an accumulate kernel is the VALU, not HBM: mixed additions per second against the measured peak of the mixed-addition routine
    alone (tools/ubench.hip: k_madd29 17.4 G/s, G1; a G2 mixed addition costs ~2.1 G1 ones).  None for any other kernel."""
    if not name.startswith("msm_accumulate") or L is None:
        return None
    wb, dg = C.c_uint32(0), C.c_uint32(0)
    lib.check(L.zk_bn254_msm_plan_info(C.c_size_t(int(n_plan)), C.c_int(1 if tables else 0), C.byref(wb), C.byref(dg)))
    madds = units_per_launch * int(dg.value)
    peak = 17.4e9 if name.endswith("g1") else 17.4e9 / 2.1
    return {"mixed_adds_per_launch": int(madds), "achieved_madd_per_s": round(madds / (per_launch_ms * 1e-3), 1), "peak_madd_per_s": peak,
            "frac": round(madds / (per_launch_ms * 1e-3) / peak, 4), "window_bits": int(wb.value),
            "peak_source": "tools/ubench.hip k_madd29, 4 waves/SIMD: 17.44 G madd/s (9,017 cycles per wave)"}


def traffic_ratio(roofline, units_per_launch, bytes_per_unit):
    """Adds the algorithmic bytes of one launch and counter traffic / algorithmic bytes (the window tables gather ~13 precomputed 64-byte rows per scalar
    where the algorithm reads the point once: the ratio is the price of that trade, stated where the roofline is)."""
    alg = float(units_per_launch) * bytes_per_unit
    roofline["algorithmic_bytes_per_launch"] = int(alg)
    roofline["traffic_over_algorithmic"] = None if not roofline.get("traffic") or alg <= 0 else round(roofline["traffic"] / alg, 2)
    roofline["valu_frac"] = (roofline.get("valu") or {}).get("frac")
    return roofline


def block_roofline(prof, steps, g1_units_per_launch, g2_units_per_launch, ntt_elems_per_launch, log_key, tables=True, L=None, lib=None, n_plan=None):
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
    roof = {"kernel": name, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic, "avg_launch_ms": round(per, 4), "launches": launches, "units_per_launch": int(units), "algorithmic_bytes_per_unit": bpu,
            "valu": valu_fraction(L, lib, name, units, per, n_plan, tables) if n_plan else None,
            "dominant_by_time": dominant_by_time(prof, steps),
            "kernel_ms_per_step": {k: round(v[1] / steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:10]}}
    return traffic_ratio(roof, units, bpu)