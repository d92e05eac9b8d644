"""Worker of tests/test_gpu_goffi.py (run as a subprocess: the shim keeps ONE SRS per process and ends the process on errors, like the reference's
log.Fatal).  argv: <json file with the inputs> -> prints one JSON object."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class GoString(C.Structure):
    _fields_ = [("p", C.c_char_p), ("n", C.c_ssize_t)]


class KeyPair(C.Structure):
    _fields_ = [("proving_key", C.c_void_p), ("verifying_key", C.c_void_p)]


def gs(s):
    b = s.encode() if isinstance(s, str) else bytes(s)
    keep.append(b)
    return GoString(b, len(b))


keep = []


def main():
    job = json.load(open(sys.argv[1]))
    Z = C.CDLL(os.path.join(ROOT, "noir_backend_using_gnark_amd", "libzkmi.so"))  # the mapping libgnark_backend.so links ($ORIGIN rpath)
    L = C.CDLL(os.path.join(ROOT, "noir_backend_using_gnark_amd", "libgnark_backend.so"))
    if os.environ.get("ZKMI_TEST_NEW_SRS_SIZE"):  # the tests' own variable: an SRS the oracle can read back in seconds (the product reads no such thing)
        assert Z.zk_export_set_new_srs_size(C.c_size_t(int(os.environ["ZKMI_TEST_NEW_SRS_SIZE"]))) == 0
    for name in ("PlonkProveWithPK", "PlonkProveWithMeta", "ProveWithMeta", "ProveWithPK"):
        getattr(L, name).restype = C.c_void_p
    for name in ("PlonkPreprocess", "Preprocess"):
        getattr(L, name).restype = KeyPair
    for name in ("PlonkVerifyWithVK", "PlonkVerifyWithMeta", "VerifyWithVK", "VerifyWithMeta"):
        getattr(L, name).restype = C.c_ubyte
    cstr = lambda p: C.string_at(p).decode()
    out = {}
    if job["what"] == "plonk":
        acir, values = job["acir"], job["values"]
        if job.get("pk"):
            pk, vk = job["pk"], job["vk"]
        else:
            kp = L.PlonkPreprocess(gs(acir), gs(json.dumps(job["random_values"])))  # the Rust side sends this one as a JSON string (main.go:66-72)
            pk, vk = cstr(kp.proving_key), cstr(kp.verifying_key)
        proof = job.get("proof") or cstr(L.PlonkProveWithPK(gs(acir), gs(values), gs(pk)))
        out = dict(pk=pk, vk=vk, proof=proof,
                   verifies=int(L.PlonkVerifyWithVK(gs(acir), gs(proof), gs(values), gs(vk))),
                   verifies_wrong_public=int(L.PlonkVerifyWithVK(gs(acir), gs(proof), gs(job["values_wrong_public"]), gs(vk))),
                   verify_with_meta=int(L.PlonkVerifyWithMeta(gs(acir), gs(values), gs(proof))))
        if not job.get("pk"):
            out["proof_with_meta"] = cstr(L.PlonkProveWithMeta(gs(acir), gs(values)))
    elif job["what"] == "groth16":
        raw = job["raw"]
        kp = L.Preprocess(gs(raw))
        pk, vk = cstr(kp.proving_key), cstr(kp.verifying_key)
        proof = cstr(L.ProveWithPK(gs(raw), gs(pk)))
        bad = proof[:192] + proof[:64]  # Krs replaced by Ar: still three valid encodings, another proof
        out = dict(pk=pk, vk=vk, proof=proof, verifies=int(L.VerifyWithVK(gs(raw), gs(proof), gs(vk))),
                   verifies_other_public=int(L.VerifyWithVK(gs(job["raw_other_public"]), gs(proof), gs(vk))),
                   proof_with_meta=cstr(L.ProveWithMeta(gs(raw))), verify_with_meta=int(L.VerifyWithMeta(gs(raw), gs(proof))))
        out["verifies_tampered"] = int(L.VerifyWithVK(gs(raw), gs(bad), gs(vk)))
    elif job["what"] == "fatal":
        L.PlonkProveWithPK(gs(job["acir"]), gs(job["values"]), gs("zz"))  # must end the process with status 1
    out["device_entries"] = int(Z.zk_device_entries(None, C.c_size_t(0)))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
