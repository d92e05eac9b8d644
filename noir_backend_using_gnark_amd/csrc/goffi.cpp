// libgnark_backend.so -- the reference's Go exports under their own names and Go's C ABI, over libzkmi: a build of the reference's Rust crate that
// links THIS library instead of the Go archive (build.rs: `cargo:rustc-link-lib=dylib=gnark_backend`) needs no other change.
//
//   PLONK  (live in the reference)      gnark_backend_ffi/main.go:24-78, declared on the Rust side at src/gnark_backend_wrapper/plonk/mod.rs:10-25
//     PlonkProveWithPK(acir, encodedValues, encodedProvingKey) *C.char       PlonkPreprocess(acir, encodedRandomValues) (*C.char, *C.char)
//     PlonkVerifyWithVK(acir, proof, publicInputs, verifyingKey) bool        PlonkVerifyWithMeta(...) bool   -- `return false` upstream (main.go:40-42)
//     PlonkProveWithMeta(acir, encodedValues) *C.char                        -- declared by the Rust side only; here: Preprocess + Prove
//   Groth16 (commented out upstream)    backend/groth16/r1cs.go:74-266, declared at src/gnark_backend_wrapper/groth16/mod.rs:14-20
//     ProveWithMeta(rawR1CS)  ProveWithPK(rawR1CS, pk)  VerifyWithMeta(rawR1CS, proof)  VerifyWithVK(rawR1CS, proof, vk)  Preprocess(rawR1CS)
//
// Go's ABI for exported functions: a `string` parameter is a GoString {const char *p; ptrdiff_t n} passed by value, `*C.char` results are malloc'ed
// (C.CString) and never freed by the Rust side, a two-value result is a struct returned by value, `bool` is one byte.
// Differences kept from libzkmi's own entry points (zk_plonk_*, zk_groth16_*): errors end the process with the message on stderr, as the reference's
// log.Fatal does; the SRS of backend/common.go:78-144 (hex(kzg.SRS.WriteTo) at <user config dir>/noir-lang/srs.hex, created with a random alpha and
// 1,000,000 points when missing) is read ONCE per process and kept resident instead of being re-read on every call (plonk.go:16,34,58).
// ZKMI_SRS_SIZE overrides the size of a newly created SRS (tests; validated: 4 .. 2^28).  Plain C++ on the C ABI: no HIP in this file.
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

#include <mutex>
#include <string>
#include <vector>

#include "zkmi.h"

extern "C" {
struct GoString {
    const char* p;
    ptrdiff_t n;
};
struct KeyPair {  // cgo's struct for a two-value result; the Rust side's #[repr(C)] KeyPair {proving_key, verifying_key}
    char* r0;
    char* r1;
};
}

namespace {

[[noreturn]] void fatal(const char* what) {
    const char* e = zk_last_error();
    fprintf(stderr, "%s: %s\n", what, e ? e : "");
    exit(1);  // log.Fatal
}
void must(int rc, const char* what) {
    if (rc != ZK_OK) fatal(what);
}
char* c_string(const std::string& s) {  // C.CString
    char* o = (char*)malloc(s.size() + 1);
    if (!o) fatal("out of memory");
    memcpy(o, s.data(), s.size());
    o[s.size()] = 0;
    return o;
}
// encodedValues reach PlonkPreprocess as a JSON string (main.go:66-72: "TODO: Fix this in the Rust backend side") and the other exports bare
std::string unquote(GoString s) {
    std::string v(s.p, (size_t)s.n);
    if (v.size() >= 2 && v.front() == '"' && v.back() == '"') v = v.substr(1, v.size() - 2);
    return v;
}
bool hex_to_bytes(const std::string& h, std::vector<uint8_t>* out) {
    if (h.size() & 1) return false;
    out->resize(h.size() / 2);
    for (size_t i = 0; i < out->size(); i++) {
        int v = 0;
        for (int k = 0; k < 2; k++) {
            const int c = h[2 * i + k], d = (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : (c >= 'A' && c <= 'F') ? c - 'A' + 10 : -1;
            if (d < 0) return false;
            v = (v << 4) | d;
        }
        (*out)[i] = (uint8_t)v;
    }
    return true;
}
// DeserializeFelts on the host for the handful of public inputs a verifier needs: u32 BE count | count x 32 B BE -> canonical big-endian elements
bool felts_from_hex(const std::string& h, std::vector<std::vector<uint8_t>>* out) {
    std::vector<uint8_t> b;
    if (!hex_to_bytes(h, &b) || b.size() < 4) return false;
    const size_t n = ((size_t)b[0] << 24) | ((size_t)b[1] << 16) | ((size_t)b[2] << 8) | b[3];
    if (b.size() != 4 + 32 * n) return false;
    out->clear();
    for (size_t i = 0; i < n; i++) out->emplace_back(b.begin() + 4 + 32 * i, b.begin() + 36 + 32 * i);
    return true;
}

// ---- the SRS of backend/common.go:78-144, once per process
struct Srs {
    std::mutex mu;
    bool ready = false;
    uint64_t handle = 0;
    zk_g2_affine g2[2];
};
Srs g_srs;
std::string srs_path() {  // os.UserConfigDir() on Linux: $XDG_CONFIG_HOME, else $HOME/.config
    const char* x = getenv("XDG_CONFIG_HOME");
    std::string dir;
    if (x && *x) dir = x;
    else {
        const char* h = getenv("HOME");
        if (!h || !*h) { fprintf(stderr, "neither $XDG_CONFIG_HOME nor $HOME are defined\n"); exit(1); }
        dir = std::string(h) + "/.config";
    }
    return dir + "/noir-lang/srs.hex";
}
// Text that hex.DecodeString accepts (LoadSRS: common.go:96-99): only then does the reference keep the file.
bool is_hex_text(const std::string& t) {
    if (t.empty() || (t.size() & 1)) return false;
    for (unsigned char c : t)
        if (!((c >= '0' && c <= '9') || (c >= 'a' && c <= 'f') || (c >= 'A' && c <= 'F'))) return false;
    return true;
}
void try_load_srs(uint64_t* handle, zk_g2_affine g2[2]) {
    std::lock_guard<std::mutex> lk(g_srs.mu);
    if (!g_srs.ready) {
        const std::string path = srs_path();
        const std::string dir = path.substr(0, path.rfind('/'));
        (void)mkdir(dir.substr(0, dir.rfind('/')).c_str(), 0755);
        (void)mkdir(dir.c_str(), 0755);
        // One process at a time decides between "load" and "generate + save": two first calls racing (the tests start workers side by side) would
        // otherwise draw two alphas, and whichever file survives invalidates the keys the other process hands out.
        const int lock_fd = open((path + ".lock").c_str(), O_CREAT | O_RDWR, 0644);
        if (lock_fd >= 0) (void)flock(lock_fd, LOCK_EX);
        struct Unlock { int fd; ~Unlock() { if (fd >= 0) { (void)flock(fd, LOCK_UN); close(fd); } } } unlock{lock_fd};
        std::string text;
        if (FILE* f = fopen(path.c_str(), "rb")) {
            char buf[1 << 16];
            size_t k;
            while ((k = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, k);
            fclose(f);
        }
        // LoadSRS fails -- and TryLoadSRS generates a new SRS -- exactly when the file cannot be read or is not hex (common.go:92-99, 129-141); it ignores what
        // ReadFrom makes of the bytes.  Here a file that IS hex is never replaced: a malformed SRS in it, or a device / memory failure while decoding it,
        // ends the process and leaves the file (and every key issued against it) alone.
        if (is_hex_text(text)) {
            size_t n = 0;
            must(zk_bn254_kzg_srs_read(text.data(), text.size(), 1, 0, &g_srs.handle, &n, g_srs.g2), "LoadSRS");
        } else {
            uint64_t a[4];
            FILE* r = fopen("/dev/urandom", "rb");
            if (!r || fread(a, 1, 32, r) != 32) { fprintf(stderr, "no randomness source\n"); exit(1); }
            fclose(r);
            a[3] &= 0x0fffffffffffffffULL;  // < 2^252 < r.  The library takes Montgomery images, and multiplication by 2^-256 permutes the field: 252 random
            zk_fr alpha;                    // bits read AS a Montgomery image are as good a secret as 252 random bits read as a value
            memcpy(&alpha, a, 32);
            const char* sz = getenv("ZKMI_SRS_SIZE");
            const long req = sz ? atol(sz) : 0;
            if (sz && (req < 4 || req > (1L << 28))) { fprintf(stderr, "ZKMI_SRS_SIZE = %s outside [4, 2^28]\n", sz); exit(1); }
            const size_t size = req ? (size_t)req : 1000000;
            void* d = nullptr;
            must(zk_dev_alloc(&d, size * 64), "NewSRS");
            must(zk_bn254_kzg_new_srs_dev(d, size, &alpha, g_srs.g2, nullptr), "NewSRS");
            must(zk_bn254_bases_register_dev(d, size, 0, &g_srs.handle), "NewSRS");
            (void)zk_dev_free(d);
            // SaveSRS: the whole text goes to a temporary file that is renamed over srs.hex, so that no reader ever sees half of it
            const size_t cap = 2 * (132 + 32 * size);
            std::string out(cap, '\0');
            size_t len = 0;
            must(zk_bn254_kzg_srs_write(g_srs.handle, g_srs.g2, 1, &out[0], cap, &len), "SaveSRS");
            const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
            bool saved = false;
            if (FILE* f = fopen(tmp.c_str(), "wb")) {
                saved = fwrite(out.data(), 1, len, f) == len;
                saved = (fflush(f) == 0) && saved;
                saved = (fclose(f) == 0) && saved;
                saved = saved && rename(tmp.c_str(), path.c_str()) == 0;
                if (!saved) (void)unlink(tmp.c_str());
            }  // like upstream (SaveSRS's error is dropped, common.go:141), a failure to save is not an error: the SRS is usable for this process
        }
        g_srs.ready = true;
    }
    *handle = g_srs.handle;
    memcpy(g2, g_srs.g2, sizeof g_srs.g2);
}

// canonical big-endian 32-byte elements -> Montgomery images (through the library's felt decoder semantics: value < r required)
bool to_mont(const std::vector<std::vector<uint8_t>>& be, std::vector<zk_fr>* out) {
    // 2^256 mod r and the modulus, for a schoolbook Montgomery conversion on the host: x * R mod r by 256 doublings
    static const uint64_t MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    auto geq = [](const uint64_t* t) {
        for (int i = 3; i >= 0; i--)
            if (t[i] != MOD[i]) return t[i] > MOD[i];
        return true;
    };
    out->resize(be.size());
    for (size_t k = 0; k < be.size(); k++) {
        uint64_t t[4];
        for (int i = 0; i < 4; i++) {
            uint64_t v = 0;
            for (int b = 0; b < 8; b++) v = (v << 8) | be[k][8 * (3 - i) + b];
            t[i] = v;
        }
        if (geq(t)) return false;
        for (int s = 0; s < 256; s++) {  // t <- 2 t mod r
            const uint64_t top = t[3] >> 63;
            for (int i = 3; i > 0; i--) t[i] = (t[i] << 1) | (t[i - 1] >> 63);
            t[0] <<= 1;
            if (top || geq(t)) {
                unsigned __int128 bw = 0;
                for (int i = 0; i < 4; i++) {
                    unsigned __int128 d = (unsigned __int128)t[i] - MOD[i] - (uint64_t)bw;
                    t[i] = (uint64_t)d;
                    bw = (d >> 64) & 1;
                }
            }
        }
        memcpy(&(*out)[k], t, 32);
    }
    return true;
}

std::string plonk_prove(GoString acir, const std::string& values, const char* pk_hex, size_t pk_len, uint64_t pk_handle, uint64_t srs) {
    std::string proof(2 * ZK_PLONK_PROOF_BYTES, '\0');
    must(zk_plonk_prove_with_pk(acir.p, (size_t)acir.n, values.data(), values.size(), ZK_ACIR_LAYOUT_REFERENCE, pk_hex, pk_len, pk_handle, srs, nullptr, &proof[0]), "PlonkProveWithPK");
    return proof;
}

}  // namespace

extern "C" {

char* PlonkProveWithPK(GoString acirJSON, GoString encodedValues, GoString encodedProvingKey) {
    uint64_t srs;
    zk_g2_affine g2[2];
    try_load_srs(&srs, g2);
    return c_string(plonk_prove(acirJSON, unquote(encodedValues), encodedProvingKey.p, (size_t)encodedProvingKey.n, 0, srs));
}

KeyPair PlonkPreprocess(GoString acirJSON, GoString encodedRandomValues) {
    uint64_t srs;
    zk_g2_affine g2[2];
    try_load_srs(&srs, g2);
    const std::string values = unquote(encodedRandomValues);
    size_t pk_len = 0, vk_len = 0;
    must(zk_plonk_preprocess(acirJSON.p, (size_t)acirJSON.n, values.data(), values.size(), ZK_ACIR_LAYOUT_REFERENCE, srs, nullptr, 0, &pk_len, nullptr, 0, &vk_len, nullptr), "PlonkPreprocess");
    std::string pk(pk_len, '\0'), vk(vk_len, '\0');
    must(zk_plonk_preprocess(acirJSON.p, (size_t)acirJSON.n, values.data(), values.size(), ZK_ACIR_LAYOUT_REFERENCE, srs, &pk[0], pk.size(), &pk_len, &vk[0], vk.size(), &vk_len, nullptr), "PlonkPreprocess");
    return KeyPair{c_string(pk), c_string(vk)};
}

char* PlonkProveWithMeta(GoString acirJSON, GoString encodedValues) {
    uint64_t srs, h = 0;
    zk_g2_affine g2[2];
    try_load_srs(&srs, g2);
    const std::string values = unquote(encodedValues);
    size_t pk_len = 0, vk_len = 0;
    must(zk_plonk_preprocess(acirJSON.p, (size_t)acirJSON.n, values.data(), values.size(), ZK_ACIR_LAYOUT_REFERENCE, srs, nullptr, 0, &pk_len, nullptr, 0, &vk_len, nullptr), "PlonkProveWithMeta");
    std::string pk(pk_len, '\0');
    must(zk_plonk_preprocess(acirJSON.p, (size_t)acirJSON.n, values.data(), values.size(), ZK_ACIR_LAYOUT_REFERENCE, srs, &pk[0], pk.size(), &pk_len, nullptr, 0, &vk_len, &h), "PlonkProveWithMeta");
    const std::string proof = plonk_prove(acirJSON, values, nullptr, 0, h, srs);
    (void)zk_bn254_plonk_pk_free(h);
    return c_string(proof);
}

unsigned char PlonkVerifyWithMeta(GoString, GoString, GoString) { return 0; }  // main.go:40-42

unsigned char PlonkVerifyWithVK(GoString acirJSON, GoString encodedProof, GoString encodedPublicInputs, GoString encodedVerifyingKey) {
    uint64_t srs;
    zk_g2_affine g2[2];
    try_load_srs(&srs, g2);
    std::vector<uint8_t> proof;
    if (!hex_to_bytes(std::string(encodedProof.p, (size_t)encodedProof.n), &proof) || proof.size() != ZK_PLONK_PROOF_BYTES) { fprintf(stderr, "DeserializeProof: not the hex of a PLONK proof\n"); exit(1); }
    // the values arrive indexed by witness (backend.rs:103: get_values_from_witness_tree over all of the circuit's variables); HandleValues keeps the
    // public ones, in witness order (common.go:45-60) -- the first n_public entries of the lowering's variable order
    std::vector<std::vector<uint8_t>> values;
    if (!felts_from_hex(unquote(encodedPublicInputs), &values)) { fprintf(stderr, "DeserializeFelts: invalid felt vector\n"); exit(1); }
    size_t n_public = 0, n_vars = 0, n_cons = 0;
    must(zk_acir_to_sparse_r1cs(acirJSON.p, (size_t)acirJSON.n, values.size(), ZK_ACIR_LAYOUT_REFERENCE, &n_public, &n_vars, &n_cons, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr), "BuildSparseR1CS");
    std::vector<uint32_t> order(n_vars ? n_vars : 1);
    must(zk_acir_to_sparse_r1cs(acirJSON.p, (size_t)acirJSON.n, values.size(), ZK_ACIR_LAYOUT_REFERENCE, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, order.data()), "BuildSparseR1CS");
    std::vector<std::vector<uint8_t>> pub_be;
    for (size_t k = 0; k < n_public; k++) pub_be.push_back(values[order[k]]);
    std::vector<zk_fr> pub;
    if (!to_mont(pub_be, &pub)) { fprintf(stderr, "DeserializeFelts: invalid fr.Element encoding\n"); exit(1); }
    int ok = 0;
    const int rc = zk_bn254_plonk_verify(proof.data(), encodedVerifyingKey.p, (size_t)encodedVerifyingKey.n, 1, g2, pub.data(), pub.size(), &ok);
    if (rc == ZK_ERR_LEN) return 0;  // plonk.Verify's "invalid witness size" is an error value upstream, i.e. `false` (plonk.go:47-50)
    must(rc, "PlonkVerifyWithVK");
    return ok ? 1 : 0;
}

// ---- Groth16 (backend/groth16/r1cs.go:74-266)
char* ProveWithMeta(GoString rawR1CS) {
    std::string proof(256, '\0');
    must(zk_groth16_prove_with_meta(rawR1CS.p, (size_t)rawR1CS.n, nullptr, nullptr, &proof[0]), "ProveWithMeta");
    return c_string(proof);
}
char* ProveWithPK(GoString rawR1CS, GoString encodedProvingKey) {
    std::string proof(256, '\0');
    must(zk_groth16_prove_with_pk(rawR1CS.p, (size_t)rawR1CS.n, encodedProvingKey.p, (size_t)encodedProvingKey.n, 0, nullptr, &proof[0]), "ProveWithPK");
    return c_string(proof);
}
KeyPair Preprocess(GoString rawR1CS) {
    size_t pk_len = 0, vk_len = 0;
    must(zk_groth16_preprocess(rawR1CS.p, (size_t)rawR1CS.n, nullptr, nullptr, 0, &pk_len, nullptr, 0, &vk_len, nullptr), "Preprocess");
    std::string pk(pk_len, '\0'), vk(vk_len, '\0');
    must(zk_groth16_preprocess(rawR1CS.p, (size_t)rawR1CS.n, nullptr, &pk[0], pk.size(), &pk_len, &vk[0], vk.size(), &vk_len, nullptr), "Preprocess");
    pk.resize(pk_len);
    vk.resize(vk_len);
    return KeyPair{c_string(pk), c_string(vk)};
}
// upstream's sketch runs a fresh Setup and verifies against ITS key (r1cs.go:145-174): no proof made elsewhere can pass -- `false`, like PlonkVerifyWithMeta
unsigned char VerifyWithMeta(GoString, GoString) { return 0; }
unsigned char VerifyWithVK(GoString rawR1CS, GoString encodedProof, GoString encodedVerifyingKey) {
    std::vector<uint8_t> proof;
    if (!hex_to_bytes(std::string(encodedProof.p, (size_t)encodedProof.n), &proof) || proof.size() != 128) { fprintf(stderr, "DeserializeProof: not the hex of a Groth16 proof\n"); exit(1); }
    uint64_t r1cs = 0;
    void* d_w = nullptr;
    size_t n_wires = 0, n_public = 0;
    must(zk_groth16_r1cs_from_raw(rawR1CS.p, (size_t)rawR1CS.n, &r1cs, &d_w, &n_wires, &n_public), "buildR1CS");
    std::vector<zk_fr> pub(n_public ? n_public : 1);
    must(zk_dev_d2h(pub.data(), d_w, n_public * 32), "buildWitnesses");  // [ONE, public...]: the public witness is everything after ONE
    (void)zk_dev_free(d_w);
    (void)zk_bn254_r1cs_free(r1cs);
    int ok = 0;
    const int rc = zk_bn254_groth16_verify(proof.data(), encodedVerifyingKey.p, (size_t)encodedVerifyingKey.n, 1, pub.data() + 1, n_public ? n_public - 1 : 0, &ok);
    if (rc == ZK_ERR_LEN) return 0;
    must(rc, "VerifyWithVK");
    return ok ? 1 : 0;
}

}  // extern "C"
