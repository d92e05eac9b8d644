"""The reference's intended Groth16 call, end to end, through libgnark_backend.so and Go's C ABI (GoString by value, C.CString results):
    Preprocess (gnark_backend_ffi/backend/groth16/r1cs.go:214-266) -> ProveWithPK (r1cs.go:107-143) -> VerifyWithVK (r1cs.go:176-212)
declared on the Rust side at src/gnark_backend_wrapper/groth16/mod.rs:14-20, on a synthetic RawR1CS (tools/synth_raw_r1cs.py; payload schema
src/gnark_backend_wrapper/groth16/acir_to_r1cs.rs:18-60) of 2^log constraints.  Worker of bench.py's `export_path_groth16` block; each mode is ONE process:

    python tools/export_bench_groth16.py make       <dir> [log_constraints] [bits]   writes raw.json (+ raw2.json: same circuit, other values); no GPU
    python tools/export_bench_groth16.py preprocess <dir>     fresh process: HIP start, Preprocess -> pk.hex, vk.hex; then a ProveWithPK in the same process
    python tools/export_bench_groth16.py prove      <dir> [warm calls]   fresh process: cold ProveWithPK, the second call, warm calls (alternating the two
                                                                         value vectors), VerifyWithVK
    python tools/export_bench_groth16.py verify     <dir>     fresh process: VerifyWithVK of the proof the prove mode left in <dir>

Every mode prints one JSON object; `phases` are the wall-clock sections the shim and the library record (zk_profile_host / prof_host), in ms."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "noir_backend_using_gnark_amd")


class GoString(C.Structure):
    _fields_ = [("p", C.c_char_p), ("n", C.c_ssize_t)]


class KeyPair(C.Structure):
    _fields_ = [("proving_key", C.c_void_p), ("verifying_key", C.c_void_p)]


def gs(b: bytes) -> GoString:
    return GoString(b, len(b))


def libs():
    Z = C.CDLL(os.path.join(PKG, "libzkmi.so"))  # the same mapping libgnark_backend.so links ($ORIGIN rpath): one library instance, one profile
    G = C.CDLL(os.path.join(PKG, "libgnark_backend.so"))
    G.ProveWithPK.restype = C.c_void_p
    G.Preprocess.restype = KeyPair
    G.VerifyWithVK.restype = C.c_ubyte
    Z.zk_profile_enable(1)
    return Z, G


def phases(Z, reset=True, kernels=None):
    """the wall-clock sections (export.*); kernels (a dict, optional) receives the eight largest kernel totals of the same interval, in ms"""
    out, ker = {}, {}
    name = C.create_string_buffer(128)
    n, ms = C.c_uint64(0), C.c_double(0)
    for i in range(Z.zk_profile_count()):
        Z.zk_profile_get(i, name, C.c_size_t(128), C.byref(n), C.byref(ms))
        k = name.value.decode()
        if k.startswith("export."):
            out[k[7:]] = round(ms.value, 3)
        elif not k.startswith("host_"):
            ker[k] = round(ms.value, 3)
    if kernels is not None:
        kernels.update(dict(sorted(ker.items(), key=lambda kv: -kv[1])[:8]))
    if reset:
        Z.zk_profile_reset()
    return out


def read(path) -> bytes:
    with open(path, "rb") as f:
        return f.read()


def main():
    mode, d = sys.argv[1], sys.argv[2]
    if mode == "make":
        from tools import synth_raw_r1cs as sr
        log_c = int(sys.argv[3]) if len(sys.argv) > 3 else 20
        bits = int(sys.argv[4]) if len(sys.argv) > 4 else 0
        t0 = time.time()
        raw, w = sr.synth(1 << (log_c - 1), 8, seed=0xE5, bits=bits)
        with open(os.path.join(d, "raw.json"), "w") as f:
            f.write(raw)
        if bits == 0:  # the same circuit text with another satisfying assignment: what the second proof of a program sends
            raw2, w2 = sr.synth(1 << (log_c - 1), 8, seed=0xE5, first=(0x1234567, 0x89abcdef))
            assert raw2[:raw2.index('"values"')] == raw[:raw.index('"values"')] and w2 != w
            with open(os.path.join(d, "raw2.json"), "w") as f:
                f.write(raw2)
        print(json.dumps({"constraints": 1 << log_c, "gates": 1 << (log_c - 1), "n_public": 8, "witnesses": len(w), "raw_bytes": len(raw), "bits_share_256": bits,
                          "make_s": round(time.time() - t0, 2)}))
        return
    raw = read(os.path.join(d, "raw.json"))
    raw2 = read(os.path.join(d, "raw2.json")) if os.path.exists(os.path.join(d, "raw2.json")) else raw
    t_start = time.perf_counter()
    Z, G = libs()
    if mode == "preprocess":
        t0 = time.perf_counter()
        kp = G.Preprocess(gs(raw))
        ms = (time.perf_counter() - t0) * 1e3
        kern = {}
        ph = phases(Z, kernels=kern)
        pk, vk = C.string_at(kp.proving_key), C.string_at(kp.verifying_key)
        with open(os.path.join(d, "pk.hex"), "wb") as f:
            f.write(pk)
        with open(os.path.join(d, "vk.hex"), "wb") as f:
            f.write(vk)
        t0 = time.perf_counter()
        proof = C.string_at(G.ProveWithPK(gs(raw), gs(pk)))
        ms_prove = (time.perf_counter() - t0) * 1e3
        ph2 = phases(Z)
        print(json.dumps({"Preprocess_ms": round(ms, 1), "phases": ph, "largest_kernels_ms": kern, "pk_text_bytes": len(pk), "vk_text_bytes": len(vk),
                          "ProveWithPK_after_preprocess_ms": round(ms_prove, 2), "phases_prove_after_preprocess": ph2,
                          "verifies": int(G.VerifyWithVK(gs(raw), gs(proof), gs(vk))), "process_s": round(time.perf_counter() - t_start, 2)}))
        return
    if mode == "prove":
        warm = int(sys.argv[3]) if len(sys.argv) > 3 else 10
        if os.environ.get("ZKMI_TOOL_BG_YIELD_MS"):  # this tool's own variable (A/B runs): how long background jobs wait for calls in flight (zk_background_set_yield_ms)
            assert Z.zk_background_set_yield_ms(C.c_int(int(os.environ["ZKMI_TOOL_BG_YIELD_MS"]))) == 0
        pk, vk = read(os.path.join(d, "pk.hex")), read(os.path.join(d, "vk.hex"))
        t0 = time.perf_counter()
        proof = C.string_at(G.ProveWithPK(gs(raw), gs(pk)))
        cold_ms = (time.perf_counter() - t0) * 1e3
        cold_kern = {}
        cold = phases(Z, kernels=cold_kern)
        t0 = time.perf_counter()
        p_second = C.string_at(G.ProveWithPK(gs(raw), gs(pk)))  # the second proving call with this key: it QUEUES the key's window tables (and the session's high-priority
        second_ms = (time.perf_counter() - t0) * 1e3            # streams) on the library's background thread and proves without them, like the first call
        second = phases(Z)
        t0 = time.perf_counter()
        idle = int(Z.zk_background_wait(C.c_int(-1)))           # ... the warm calls below are measured with the tables in place (the steady state)
        bg_ms = (time.perf_counter() - t0) * 1e3
        background = phases(Z)
        with open(os.path.join(d, "proof.hex"), "wb") as f:
            f.write(proof)
        t0 = time.perf_counter()
        for k in range(warm):
            p2 = C.string_at(G.ProveWithPK(gs(raw2 if k & 1 == 0 else raw), gs(pk)))  # ends on raw when warm is even
        warm_ms = (time.perf_counter() - t0) * 1e3 / warm
        wph = {k: round(v / warm, 3) for k, v in phases(Z).items()}
        t0 = time.perf_counter()
        ok = int(G.VerifyWithVK(gs(raw), gs(proof), gs(vk)))
        ver_ms = (time.perf_counter() - t0) * 1e3
        ok2 = int(G.VerifyWithVK(gs(raw if warm % 2 == 0 else raw2), gs(p2), gs(vk)))
        ok_second = int(G.VerifyWithVK(gs(raw), gs(p_second), gs(vk)))  # the proof made while the tables were being built
        # another public input must be rejected: witness 1 is public; its last hex digit sits at a fixed offset of the values string
        a = raw.index(b'"values":"') + len(b'"values":"')
        bad = bytearray(raw)
        bad[a + 8 + 63] = ord("1") if bad[a + 8 + 63] != ord("1") else ord("2")
        rej = int(G.VerifyWithVK(gs(bytes(bad)), gs(proof), gs(vk)))
        nc, nk, by = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        if hasattr(Z, "zk_export_cache_info"):
            Z.zk_export_cache_info(C.byref(nc), C.byref(nk), C.byref(by))
        prove = wph.get("groth16_prove_r1cs", 0.0)
        print(json.dumps({"cold_ProveWithPK_ms": round(cold_ms, 1), "cold_phases": cold, "cold_largest_kernels_ms": cold_kern, "second_ProveWithPK_ms": round(second_ms, 2), "second_phases": second,
                          "background_after_second_ms": round(bg_ms, 2), "background_idle": idle, "background_phases": background, "second_proof_verifies": ok_second,
                          "warm_ProveWithPK_ms": round(warm_ms, 3), "warm_calls": warm, "warm_phases_per_call": wph,
                          "zk_bn254_groth16_prove_r1cs_ms": prove, "warm_over_prove": round(warm_ms / prove, 3) if prove else None,
                          "VerifyWithVK_ms": round(ver_ms, 2), "verifies": ok, "warm_proof_verifies": ok2, "wrong_public_input_rejected": int(rej == 0),
                          "text_bytes_in": {"raw_r1cs": len(raw), "pk": len(pk)}, "resident": {"circuits": nc.value, "keys": nk.value, "bytes": by.value},
                          "process_s": round(time.perf_counter() - t_start, 2)}))
        return
    if mode == "verify":
        vk, proof = read(os.path.join(d, "vk.hex")), read(os.path.join(d, "proof.hex"))
        t0 = time.perf_counter()
        ok = int(G.VerifyWithVK(gs(raw), gs(proof), gs(vk)))
        cold_ms = (time.perf_counter() - t0) * 1e3
        ph = phases(Z)
        t0 = time.perf_counter()
        ok2 = int(G.VerifyWithVK(gs(raw), gs(proof), gs(vk)))
        warm_ms = (time.perf_counter() - t0) * 1e3
        print(json.dumps({"cold_VerifyWithVK_ms": round(cold_ms, 1), "cold_phases": ph, "second_VerifyWithVK_ms": round(warm_ms, 2), "verifies": ok & ok2,
                          "device_entries": int(Z.zk_device_entries(None, C.c_size_t(0))), "process_s": round(time.perf_counter() - t_start, 2)}))
        return
    raise SystemExit("unknown mode " + mode)


if __name__ == "__main__":
    main()
