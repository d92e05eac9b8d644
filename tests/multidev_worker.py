"""Worker of tests/test_gpu_multidev.py (a subprocess: zk_init_devices changes process-wide state).  One process, several device ENTRIES -- this pool has
one GPU per box, so the SAME device is listed 8 times (virtual devices: own streams, workspaces, tables; "peer copies" are device-local) and, when the box
has more than one GPU, the real ones too.  Every multi-entry result must equal the single-entry bytes.  argv: <mode> ; prints one JSON object."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import noir_backend_using_gnark_amd as zk  # noqa: E402
from noir_backend_using_gnark_amd import _lib, bn254 as zb, kzg  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (the checker)

L = _lib.lib()
MONT = zk.MultiExpConfig(scalars_mont=True)


def init(devs):
    arr = (C.c_int * len(devs))(*devs)
    _lib.check(L.zk_init_devices(arr, C.c_size_t(len(devs))))
    got = (C.c_int * 64)()
    n = L.zk_device_entries(got, C.c_size_t(64))
    assert n == len(devs) and list(got[:n]) == list(devs)
    _lib.check(L.zk_set_default_devices(C.c_uint32(0)))  # tests name their masks; the default stays "this thread's entry"


def mask_of(k, first=0):
    return sum(1 << (first + i) for i in range(k))


def main():
    mode = sys.argv[1]
    real = int(L.zk_device_count())
    devs = [0] * 8 if mode != "real_peers" else list(range(real))
    init(devs)
    out = {"entries": len(devs), "real_devices": real}
    if mode == "msm":
        n = 5000
        pts, sc = orc.g1_gen_points(11, n), orc.rand_fr(12, n)
        p2 = orc.g2_gen_points(13, 700)
        want1, want2 = orc.g1_msm(pts, sc), orc.g2_msm(p2, sc[:700])
        res = {}
        for k in (1, 2, 3, 4, 8):
            cfg = zk.MultiExpConfig(scalars_mont=True, device_mask=mask_of(k))
            res["g1_%d" % k] = bool((zk.g1_multi_exp(pts, sc, config=cfg) == want1).all())
            res["g2_%d" % k] = bool((zk.g2_multi_exp(p2, sc[:700], config=cfg) == want2).all())
        res["single_bit_entry5"] = bool((zk.g1_multi_exp(pts, sc, config=zk.MultiExpConfig(scalars_mont=True, device_mask=1 << 5)) == want1).all())
        res["more_entries_than_points"] = bool((zk.g1_multi_exp(pts[:3], sc[:3], config=zk.MultiExpConfig(scalars_mont=True, device_mask=0xff)) == orc.g1_msm(pts[:3], sc[:3])).all())
        try:
            zk.g1_multi_exp(pts, sc, config=zk.MultiExpConfig(scalars_mont=True, device_mask=1 << 9))
            res["bad_mask_refused"] = False
        except Exception:
            res["bad_mask_refused"] = True
        out.update(res)
    elif mode == "bases":
        # composite resident bases: the process default names 2 (then 4) entries, 2^18 generated points stay resident by range; commits with host and with
        # device scalars, with offsets that start in the middle of an entry's range, against the single-entry handle of the same points
        n = 1 << 18
        dp, ds = _lib.DeviceBuffer(n * 64), _lib.DeviceBuffer(n * 32)
        _lib.check(L.zk_bn254_g1_generate_dev(C.c_void_p(dp.ptr), C.c_size_t(n), C.c_uint64(0x51), None))
        _lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(ds.ptr), C.c_size_t(n), C.c_uint64(0x52), C.c_int(1), C.c_int(0), None))
        sc = ds.to_numpy(np.uint64, (n, 4))
        single = zb.ResidentBases(dp, n=n)
        res = {}
        for k in (2, 4):
            _lib.check(L.zk_set_default_devices(C.c_uint32(mask_of(k))))
            comp = zb.ResidentBases(dp, n=n)
            res["composite_handle_%d" % k] = (comp.handle.value >> 56) == 0xff
            for off, cnt in ((0, n), (0, 1000), (n // k - 7, 5000), (n - 3000, 3000), (12345, n // 2)):
                a = single.multi_exp(sc[:cnt], MONT, offset=off)
                res["host_%d_%d_%d" % (k, off, cnt)] = bool((comp.multi_exp(sc[:cnt], MONT, offset=off) == a).all())
                res["dev_%d_%d_%d" % (k, off, cnt)] = bool((comp.multi_exp_dev(ds, cnt, MONT, offset=off) == a).all())
            try:
                comp.multi_exp(sc[:10], MONT, offset=n - 5)
                res["overrun_refused_%d" % k] = False
            except ValueError:
                res["overrun_refused_%d" % k] = True
            comp.free()
            # registered WITHOUT window tables, then zk_bn254_bases_build_table on the composite handle (every entry builds its range's): same commitments
            late = zb.ResidentBases(dp, n=n, table_window_bits=-1)
            a = single.multi_exp(sc[:70000], MONT, offset=999)
            before = bool((late.multi_exp(sc[:70000], MONT, offset=999) == a).all())
            late.build_table()
            late.build_table()  # a second call finds them
            res["tables_built_later_%d" % k] = before and bool((late.multi_exp(sc[:70000], MONT, offset=999) == a).all()) and bool(
                (late.multi_exp_dev(ds, n, MONT) == single.multi_exp_dev(ds, n, MONT)).all())
            late.free()
        _lib.check(L.zk_set_default_devices(C.c_uint32(0)))
        out.update(res)
    elif mode == "groth16":
        import bench
        log_n = 12
        inst = bench.Instance(L, _lib, zk, log_n, 0, bench.N_PUBLIC, 1, True)
        want = zk.prove(inst.pk, inst.d_a, inst.d_b, inst.d_c, inst.d_w, inst.r, inst.s, n_constraints=inst.N, on_device=True)
        cpu, _, _ = bench.oracle_proof(inst, log_n)
        res = {"single_equals_oracle": want == cpu}
        ha, hb, hc, hw = (d.to_numpy(np.uint64, (inst.N, 4)) for d in (inst.d_a, inst.d_b, inst.d_c, inst.d_w))
        for k in (2, 4, 8):
            pk = zk.ProvingKey(log_n, inst.N, bench.N_PUBLIC, inst.small["alpha"], inst.small["beta"], inst.small["delta"], inst.g1_a, inst.g1_b,
                               inst.g1_k.ptr + bench.N_PUBLIC * 64, inst.g1_z, inst.small2["beta"], inst.small2["delta"], inst.g2_b, bases_on_device=True,
                               device_mask=mask_of(k))
            res["composite_handle_%d" % k] = (pk.handle.value >> 56) == 0xff and pk.info()["n_wires"] == inst.N
            res["dev_inputs_%d" % k] = zk.prove(pk, inst.d_a, inst.d_b, inst.d_c, inst.d_w, inst.r, inst.s, n_constraints=inst.N, on_device=True) == want
            res["host_inputs_%d" % k] = zk.prove(pk, ha, hb, hc, hw, inst.r, inst.s) == want
            res["again_%d" % k] = zk.prove(pk, ha, hb, hc, hw, inst.r, inst.s) == want
            pk.free()
        # host-resident bases (gnark's slices), fewer constraints than the domain, entries 2..5
        g = lambda b, w: b.to_numpy(np.uint64, (inst.N, w))
        pk = zk.ProvingKey(log_n, inst.N, bench.N_PUBLIC, inst.small["alpha"], inst.small["beta"], inst.small["delta"], g(inst.g1_a, 8), g(inst.g1_b, 8),
                           g(inst.g1_k, 8)[bench.N_PUBLIC:], g(inst.g1_z, 8), inst.small2["beta"], inst.small2["delta"], g(inst.g2_b, 16), device_mask=mask_of(4, 2))
        nc = inst.N - 37
        single_nc = zk.prove(inst.pk, ha[:nc], hb[:nc], hc[:nc], hw, inst.r, inst.s)
        res["host_bases_short_abc_entries_2_5"] = zk.prove(pk, ha[:nc], hb[:nc], hc[:nc], hw, inst.r, inst.s) == single_nc
        pk.free()
        try:
            zk.ProvingKey(log_n, inst.N, bench.N_PUBLIC, inst.small["alpha"], inst.small["beta"], inst.small["delta"], inst.g1_a, inst.g1_b,
                          inst.g1_k.ptr + bench.N_PUBLIC * 64, inst.g1_z, inst.small2["beta"], inst.small2["delta"], inst.g2_b, bases_on_device=True, device_mask=mask_of(3))
            res["three_entries_refused"] = False
        except Exception:
            res["three_entries_refused"] = True
        inst.free()
        out.update(res)
    elif mode == "ntt":
        res = {}
        for log_n in (8, 13):
            x = orc.rand_fr(0x77 + log_n, 1 << log_n)
            for inverse in (0, 1):
                for dec in (0, 1):
                    for coset in (0, 1):
                        want = orc.fr_ntt(x, bool(inverse), dec, bool(coset))
                        for k in (2, 4, 8):
                            if log_n < 2 * (k.bit_length() - 1) + 2:
                                continue
                            y = x.copy()
                            _lib.check(L.zk_bn254_ntt_devices(_lib.vp(y), C.c_uint32(log_n), C.c_int(inverse), C.c_int(dec), C.c_int(coset), C.c_uint32(mask_of(k))))
                            res["n%d_inv%d_dec%d_coset%d_k%d" % (log_n, inverse, dec, coset, k)] = bool((y == want).all())
        out.update(res)
    elif mode == "entries":
        # handles carry their entry: an object made on entry 3 is used from a thread on entry 0; zk_set_entry moves the handle-less calls
        n = 3000
        pts, sc = orc.g1_gen_points(21, n), orc.rand_fr(22, n)
        want = orc.g1_msm(pts, sc)
        _lib.check(L.zk_set_entry(3))
        rb = zb.ResidentBases(pts)
        d = _lib.DeviceBuffer.from_numpy(sc)
        res = {"handle_entry": rb.handle.value >> 56, "on_entry_3": bool((rb.multi_exp(sc, MONT) == want).all())}
        x = orc.rand_fr(23, 1 << 10)
        y = x.copy()
        zk.Domain(1 << 10).fft(y, zk.DIF)
        res["ntt_on_entry_3"] = bool((y == orc.fr_ntt(x, False, orc.DIF)).all())
        _lib.check(L.zk_set_entry(0))
        res["from_entry_0"] = bool((rb.multi_exp(sc, MONT) == want).all()) and bool((rb.multi_exp_dev(d, n, MONT) == want).all())
        rb.free()
        res["unknown_entry_refused"] = L.zk_set_entry(9) != 0
        out.update(res)
    elif mode == "plonk":
        # the reference's live prover against an SRS that is spread over 2 entries (what libgnark_backend.so gets when the process names several GPUs):
        # the golden proof bytes with pinned blinders
        from noir_backend_using_gnark_amd import frontend as fe
        from oracle import bn254_ref as ref, plonk_ref as pl
        e = json.load(open(os.path.join(ROOT, "tests", "golden", "plonk_golden.json")))[1]
        h2i = lambda h: int(h, 16)
        M = pl.ints_to_mont_np
        _lib.check(L.zk_set_default_devices(C.c_uint32(mask_of(2))))
        srs = kzg.new_srs(1 << 18, M([h2i(e["srs_alpha"])])[0])     # the first points are the golden SRS's (powers of the same alpha)
        _lib.check(L.zk_set_default_devices(C.c_uint32(0)))
        values = [h2i(v) for v in e["values"]]
        enc = ref.felts_wire(values).hex()
        bl = M([h2i(v) for v in e["blinders"]])
        pk_hex, vk_hex = fe.plonk_preprocess(json.dumps(e["acir"]), enc, srs)
        out.update({"composite_srs": (srs.handle.value >> 56) == 0xff, "key_text_equals_golden": pk_hex == e["pk_hex"],
                    "proof_equals_golden": fe.plonk_prove_with_pk(json.dumps(e["acir"]), enc, pk_hex, srs, blinders=bl) == e["proof"]})
        srs.free()
    elif mode == "real_peers":
        if real < 2:
            out["skipped"] = "one GPU on this box"
        else:
            pts, sc = orc.g1_gen_points(31, 4096), orc.rand_fr(32, 4096)
            want = orc.g1_msm(pts, sc)
            out["msm_all_real_devices"] = bool((zk.g1_multi_exp(pts, sc, config=zk.MultiExpConfig(scalars_mont=True, device_mask=mask_of(real))) == want).all())
            k = 1 << (real.bit_length() - 1)  # the largest power of two of real GPUs: Groth16 and the transform shard over 2, 4 or 8
            k = min(k, 8)
            import bench
            log_n = 12
            inst = bench.Instance(L, _lib, zk, log_n, 0, bench.N_PUBLIC, 1, True)
            single = zk.prove(inst.pk, inst.d_a, inst.d_b, inst.d_c, inst.d_w, inst.r, inst.s, n_constraints=inst.N, on_device=True)
            pk = zk.ProvingKey(log_n, inst.N, bench.N_PUBLIC, inst.small["alpha"], inst.small["beta"], inst.small["delta"], inst.g1_a, inst.g1_b,
                               inst.g1_k.ptr + bench.N_PUBLIC * 64, inst.g1_z, inst.small2["beta"], inst.small2["delta"], inst.g2_b, bases_on_device=True, device_mask=mask_of(k))
            out["groth16_%d_real_devices_device_inputs" % k] = zk.prove(pk, inst.d_a, inst.d_b, inst.d_c, inst.d_w, inst.r, inst.s, n_constraints=inst.N, on_device=True) == single
            ha, hb, hc, hw = (d.to_numpy(np.uint64, (inst.N, 4)) for d in (inst.d_a, inst.d_b, inst.d_c, inst.d_w))
            out["groth16_%d_real_devices_host_inputs" % k] = zk.prove(pk, ha, hb, hc, hw, inst.r, inst.s) == single
            pk.free()
            x = orc.rand_fr(0x99, 1 << 13)
            y = x.copy()
            _lib.check(L.zk_bn254_ntt_devices(_lib.vp(y), C.c_uint32(13), C.c_int(1), C.c_int(1), C.c_int(1), C.c_uint32(mask_of(k))))
            out["ntt_%d_real_devices" % k] = bool((y == orc.fr_ntt(x, True, 1, True)).all())
            n = 1 << 18
            dp, ds = _lib.DeviceBuffer(n * 64), _lib.DeviceBuffer(n * 32)
            _lib.check(L.zk_bn254_g1_generate_dev(C.c_void_p(dp.ptr), C.c_size_t(n), C.c_uint64(0x51), None))
            _lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(ds.ptr), C.c_size_t(n), C.c_uint64(0x52), C.c_int(1), C.c_int(0), None))
            one = zb.ResidentBases(dp, n=n)
            _lib.check(L.zk_set_default_devices(C.c_uint32(mask_of(min(real, 4)))))
            comp = zb.ResidentBases(dp, n=n)
            _lib.check(L.zk_set_default_devices(C.c_uint32(0)))
            out["composite_bases_device_scalars_over_peers"] = bool((comp.multi_exp_dev(ds, n, MONT) == one.multi_exp_dev(ds, n, MONT)).all())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
