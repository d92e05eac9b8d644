"""The reference's live call, end to end, through libgnark_backend.so and Go's C ABI (GoString by value, C.CString results):
    PlonkPreprocess (gnark_backend_ffi/main.go:58-78) -> PlonkProveWithPK (main.go:24-37) -> PlonkVerifyWithVK (main.go:44-56)
on a synthetic ACIR of 2^19 - 8 arithmetic opcodes + 8 public inputs (tools/synth_acir.py: the largest domain under the reference's 1,000,000-point
SRS, backend/common.go:137), reference variable layout.  Worker of bench.py's `export_path` block; each mode is ONE process, as nargo runs them:

    python tools/export_bench.py make      <dir> [log_gates]     writes acir.json, values.hex (no GPU)
    python tools/export_bench.py preprocess <dir>                 fresh process: HIP start, SRS generate + save, PlonkPreprocess -> pk.hex, vk.hex
    python tools/export_bench.py prove      <dir> [warm calls]    fresh process: HIP start, SRS load, cold PlonkProveWithPK, the second call, then warm calls, PlonkVerifyWithVK
    python tools/export_bench.py verify     <dir>                 fresh process: PlonkVerifyWithVK of the proof the prove mode left in <dir>

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
    G.PlonkProveWithPK.restype = C.c_void_p
    G.PlonkPreprocess.restype = KeyPair
    G.PlonkVerifyWithVK.restype = C.c_ubyte
    Z.zk_profile_enable(1)
    if os.environ.get("ZKMI_TEST_NEW_SRS_SIZE"):  # a test asks for an SRS its oracle can read back in seconds (default: the reference's 1,000,000 points)
        assert Z.zk_export_set_new_srs_size(C.c_size_t(int(os.environ["ZKMI_TEST_NEW_SRS_SIZE"]))) == 0
    return Z, G


def phases(Z, reset=True):
    out = {}
    name = C.create_string_buffer(128)
    n, ms = C.c_uint64(0), C.c_double(0)
    for i in range(Z.zk_profile_count()):
        Z.zk_profile_get(i, name, C.c_size_t(128), C.byref(n), C.byref(ms))
        k = name.value.decode()
        if k.startswith("export."):
            out[k[7:]] = round(ms.value, 3)
    if reset:
        Z.zk_profile_reset()
    return out


def read(path) -> bytes:
    with open(path, "rb") as f:
        return f.read()


def main():
    mode, d = sys.argv[1], sys.argv[2]
    os.environ["XDG_CONFIG_HOME"] = os.path.join(d, "cfg")  # srs.hex of this run lives and dies with the directory
    os.makedirs(os.path.join(d, "cfg"), exist_ok=True)
    if mode == "make":
        from tools import synth_acir
        log_g = int(sys.argv[3]) if len(sys.argv) > 3 else 19
        n_public = 8
        t0 = time.time()
        acir, w = synth_acir.synth((1 << log_g) - n_public, n_public, seed=0xE4)
        with open(os.path.join(d, "acir.json"), "w") as f:
            f.write(acir)
        with open(os.path.join(d, "values.hex"), "w") as f:
            f.write(synth_acir.felts_wire_hex(w))
        print(json.dumps({"opcodes": (1 << log_g) - n_public, "n_public": n_public, "witnesses": len(w), "acir_bytes": len(acir), "make_s": round(time.time() - t0, 2)}))
        return
    acir, values = read(os.path.join(d, "acir.json")), read(os.path.join(d, "values.hex"))
    t_start = time.perf_counter()
    Z, G = libs()
    if mode == "preprocess":
        t0 = time.perf_counter()
        kp = G.PlonkPreprocess(gs(acir), gs(b'"' + values + b'"'))  # the Rust side sends this one as a JSON string (main.go:66-72)
        ms = (time.perf_counter() - t0) * 1e3
        ph = phases(Z)
        pk, vk = C.string_at(kp.proving_key), C.string_at(kp.verifying_key)
        with open(os.path.join(d, "pk.hex"), "wb") as f:
            f.write(pk)
        with open(os.path.join(d, "vk.hex"), "wb") as f:
            f.write(vk)
        # the process that made the key also proves warm: its key is resident under the content key of the text it returned
        t0 = time.perf_counter()
        proof = C.string_at(G.PlonkProveWithPK(gs(acir), gs(values), gs(pk)))
        ms_prove = (time.perf_counter() - t0) * 1e3
        ph2 = phases(Z)
        print(json.dumps({"PlonkPreprocess_ms": round(ms, 1), "phases": ph, "pk_text_bytes": len(pk), "vk_text_bytes": len(vk),
                          "PlonkProveWithPK_after_preprocess_ms": round(ms_prove, 2), "phases_prove_after_preprocess": ph2,
                          "verifies": int(G.PlonkVerifyWithVK(gs(acir), gs(proof), gs(values), gs(vk))), "process_s": round(time.perf_counter() - t_start, 2)}))
        return
    if mode == "prove":
        warm = int(sys.argv[3]) if len(sys.argv) > 3 else 10
        if os.environ.get("ZKMI_TOOL_BG_YIELD_MS"):  # this tool's own variable (A/B runs): how long background jobs wait for calls in flight (zk_background_set_yield_ms)
            assert Z.zk_background_set_yield_ms(C.c_int(int(os.environ["ZKMI_TOOL_BG_YIELD_MS"]))) == 0
        pk, vk = read(os.path.join(d, "pk.hex")), read(os.path.join(d, "vk.hex"))
        t0 = time.perf_counter()
        proof = C.string_at(G.PlonkProveWithPK(gs(acir), gs(values), gs(pk)))
        cold_ms = (time.perf_counter() - t0) * 1e3
        cold = phases(Z)
        t0 = time.perf_counter()
        p_second = C.string_at(G.PlonkProveWithPK(gs(acir), gs(values), gs(pk)))  # the second proving call of the process: it QUEUES the SRS's window tables on the
        second_ms = (time.perf_counter() - t0) * 1e3                             # library's background thread (csrc/goffi.cpp srs_for_repeat_use) and commits without them
        second = phases(Z)
        t0 = time.perf_counter()
        idle = int(Z.zk_background_wait(C.c_int(-1)))                            # ... the warm calls below are measured with the tables in place
        bg_ms = (time.perf_counter() - t0) * 1e3
        background = phases(Z)
        with open(os.path.join(d, "proof.hex"), "wb") as f:
            f.write(proof)
        t0 = time.perf_counter()
        for _ in range(warm):
            p2 = C.string_at(G.PlonkProveWithPK(gs(acir), gs(values), gs(pk)))
        warm_ms = (time.perf_counter() - t0) * 1e3 / warm
        wph = {k: round(v / warm, 3) for k, v in phases(Z).items()}
        t0 = time.perf_counter()
        ok = int(G.PlonkVerifyWithVK(gs(acir), gs(proof), gs(values), gs(vk)))
        ver_ms = (time.perf_counter() - t0) * 1e3
        ok2 = int(G.PlonkVerifyWithVK(gs(acir), gs(p2), gs(values), gs(vk)))
        ok_second = int(G.PlonkVerifyWithVK(gs(acir), gs(p_second), gs(values), gs(vk)))  # the proof made while the tables were being built
        bad = bytearray(values)
        bad[8 + 63] = ord("1") if bad[8 + 63] != ord("1") else ord("2")  # witness 1 is public: another public input must be rejected
        rej = int(G.PlonkVerifyWithVK(gs(acir), gs(proof), gs(bytes(bad)), gs(vk)))
        nc, nk, by = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        Z.zk_export_cache_info(C.byref(nc), C.byref(nk), C.byref(by))
        prove = wph.get("plonk_prove", 0.0)
        print(json.dumps({"cold_PlonkProveWithPK_ms": round(cold_ms, 1), "cold_phases": cold, "second_PlonkProveWithPK_ms": round(second_ms, 2), "second_phases": second,
                          "background_after_second_ms": round(bg_ms, 2), "background_idle": idle, "background_phases": background, "second_proof_verifies": ok_second,
                          "warm_PlonkProveWithPK_ms": round(warm_ms, 3), "warm_calls": warm,
                          "warm_phases_per_call": wph, "zk_bn254_plonk_prove_ms": prove, "warm_over_prove": round(warm_ms / prove, 3) if prove else None,
                          "PlonkVerifyWithVK_ms": round(ver_ms, 2), "verifies": ok, "warm_proof_verifies": ok2, "wrong_public_input_rejected": int(rej == 0),
                          "text_bytes_in": {"acir": len(acir), "values": len(values), "pk": len(pk)}, "resident": {"circuits": nc.value, "keys": nk.value, "bytes": by.value},
                          "process_s": round(time.perf_counter() - t_start, 2)}))
        return
    if mode == "verify":  # a process that only verifies (nargo verify): the SRS's two G2 points come from the file's header on the host -- no HIP runtime, no device
        vk, proof = read(os.path.join(d, "vk.hex")), read(os.path.join(d, "proof.hex"))
        t0 = time.perf_counter()
        ok = int(G.PlonkVerifyWithVK(gs(acir), gs(proof), gs(values), gs(vk)))
        cold_ms = (time.perf_counter() - t0) * 1e3
        ph = phases(Z)
        t0 = time.perf_counter()
        ok2 = int(G.PlonkVerifyWithVK(gs(acir), gs(proof), gs(values), gs(vk)))
        warm_ms = (time.perf_counter() - t0) * 1e3
        print(json.dumps({"cold_PlonkVerifyWithVK_ms": round(cold_ms, 1), "cold_phases": ph, "second_PlonkVerifyWithVK_ms": round(warm_ms, 2), "verifies": ok & ok2,
                          "hip_runtime_started": "hip_init" in ph, "device_entries": int(Z.zk_device_entries(None, C.c_size_t(0))), "process_s": round(time.perf_counter() - t_start, 2)}))
        return
    raise SystemExit("unknown mode " + mode)


if __name__ == "__main__":
    main()
