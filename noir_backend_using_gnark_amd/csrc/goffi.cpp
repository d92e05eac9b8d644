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
// Devices: ONE -- the first visible GPU -- unless ZKMI_DEVICES says otherwise ("all", or a comma-separated list of HIP device ordinals; validated; read once, before
// anything touches a device): several GPUs in one process (csrc/multidev.hip) have only ever run on virtual entries, so they are opt-in.
// The size of an SRS this process CREATES is libzkmi's zk_export_new_srs_size() (1,000,000 unless a test called the setter): this library exports the reference's
// ten names and nothing else, and reads no variable that could change a result.  Plain C++ on the C ABI: no HIP here.
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <future>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "zkmi.h"
#include "text_host.hpp"
#include "acir_host.hpp"

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
// the text helpers (unquote, hex_to_bytes, felts_from_hex, be_to_mont) live in text_host.hpp with the other readers of untrusted bytes: host only,
// sanitizer- and mutation-tested on the CPU (tests/cpp/parser_fuzz.cpp)
using zkmi::be_to_mont;
using zkmi::felts_from_hex;
using zkmi::hex_to_bytes;
struct View {  // a GoString's payload without its JSON quotes (main.go:66-72), borrowed for the call
    const char* p;
    size_t n;
};
View unquoted(GoString s) {
    View v;
    zkmi::unquote(s.p, (size_t)s.n, &v.p, &v.n);
    return v;
}
struct Lap {  // wall-clock sections of the shim, reported beside the library's own when profiling is on (bench.py `export_path`)
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char* name) {
        const auto t1 = std::chrono::steady_clock::now();
        (void)zk_profile_host(name, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// ---- which GPUs this process uses.  The reference is one process per nargo command and knows nothing of devices: the default is ONE device entry (HIP device
// 0 of those visible), whatever the node holds -- starting a HIP context per GPU costs each one-shot call its start-up time over again, and concurrent nargo
// processes would contend for every GPU.  ZKMI_DEVICES=all | d0,d1,... opts in to the in-process multi-GPU path (the SRS kept by range on the listed devices, every
// commitment one partial per GPU); anything else ends the process with a message (a typo must not silently mean "one GPU").
std::vector<int> shim_devices() {
    const char* v = getenv("ZKMI_DEVICES");
    std::vector<int> devs;
    if (!v || !*v) { devs.push_back(0); return devs; }
    if (!strcmp(v, "all")) return devs;  // empty = every visible device (zk_init_devices(NULL, 0))
    const int visible = zk_device_count();
    const char* p = v;
    while (*p) {
        char* e = nullptr;
        const long d = strtol(p, &e, 10);
        if (e == p || d < 0 || d >= visible || (*e && *e != ',')) { fprintf(stderr, "ZKMI_DEVICES = \"%s\": expected \"all\" or a comma-separated list of device ordinals below %d\n", v, visible); exit(1); }
        devs.push_back((int)d);
        p = *e ? e + 1 : e;
    }
    if (devs.empty() || devs.size() > 8) { fprintf(stderr, "ZKMI_DEVICES = \"%s\": one to eight devices\n", v); exit(1); }
    return devs;
}
// every proving export is one "call in flight" for the library's background thread (window tables, streams): jobs queued by a call start after it (csrc/ctx.hip)
struct InFlight {
    InFlight() { zk_background_hold(1); }
    ~InFlight() { zk_background_hold(-1); }
};
void start_devices(int warm_streams) {
    must(zk_init_flags(ZK_INIT_LEAN_STREAMS), "zk_init_flags");  // a process behind these exports typically makes ONE call: no stream it will not use
    const std::vector<int> devs = shim_devices();
    must(zk_init_devices(devs.empty() ? nullptr : devs.data(), devs.size()), "zk_init_devices");
    if (warm_streams) must(zk_warm_streams(warm_streams), "zk_warm_streams");
}

// ---- the SRS of backend/common.go:78-144, once per process
struct Srs {
    std::mutex mu;
    bool ready = false;     // the G1 bases are resident (handle) and g2 is set
    bool g2_only = false;   // g2 is set from the file's header and nothing is on a device (a process that has only verified so far)
    bool tables = false;    // the bases have their window tables
    int uses = 0;           // exports served that commit against the SRS (Preprocess, Prove*)
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
// Text that hex.DecodeString accepts (LoadSRS: common.go:96-99): only then does the reference keep the file.  64 MB for the reference's 1,000,000 points.
bool is_hex_text(const std::string& t) { return !t.empty() && !(t.size() & 1) && zkmi::all_hex(t.data(), t.size()); }
std::string read_srs_text(const std::string& path) {  // 64 MB of text for the reference's 1,000,000 points: sized once, read in one piece
    std::string text;
    if (FILE* f = fopen(path.c_str(), "rb")) {
        struct stat sb;
        if (fstat(fileno(f), &sb) == 0 && sb.st_size > 0) {
            text.resize((size_t)sb.st_size);
            const size_t got = fread(&text[0], 1, text.size(), f);
            text.resize(got);
        }
        fclose(f);
    }
    return text;
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
        // The HIP runtime starts on a thread of its own (0.07-0.2 s) and then creates the streams the first proof will take (the process's first one: 40-160 ms,
        // 10 ms each after) while this thread reads the file.  One device entry unless ZKMI_DEVICES opts in to more (shim_devices).
        std::thread starter([] { start_devices(3); });
        Lap lap;
        const std::string text = read_srs_text(path);
        // LoadSRS fails -- and TryLoadSRS generates a new SRS -- exactly when the file cannot be read or is not hex (common.go:92-99, 129-141); it ignores what
        // ReadFrom makes of the bytes.  Here a file that IS hex is never replaced: a malformed SRS in it, or a device / memory failure while decoding it,
        // ends the process and leaves the file (and every key issued against it) alone.
        const bool usable = is_hex_text(text);
        lap.lap("export.srs_file_read");
        starter.join();
        lap.lap("export.hip_start_wait");  // what the runtime and the first streams still took once the file was in memory
        // No window tables yet: for 1,000,000 points they take 17 ms to build and save a 2^19-gate proof 1.4 ms -- a process that makes one proof (nargo prove)
        // is better off without; srs_for_repeat_use builds them when a second proving call arrives.
        if (usable) {
            size_t n = 0;
            must(zk_bn254_kzg_srs_read(text.data(), text.size(), 1, -1, &g_srs.handle, &n, g_srs.g2), "LoadSRS");
            lap.lap("export.srs_decode");
        } else {
            uint64_t a[4];
            FILE* r = fopen("/dev/urandom", "rb");
            if (!r || fread(a, 1, 32, r) != 32) { fprintf(stderr, "no randomness source\n"); exit(1); }
            fclose(r);
            a[3] &= 0x0fffffffffffffffULL;  // < 2^252 < r.  The library takes Montgomery images, and multiplication by 2^-256 permutes the field: 252 random
            zk_fr alpha;                    // bits read AS a Montgomery image are as good a secret as 252 random bits read as a value
            memcpy(&alpha, a, 32);
            const size_t size = zk_export_new_srs_size();  // the reference's 1,000,000 points (backend/common.go:137) unless a test asked for fewer
            void* d = nullptr;
            must(zk_dev_alloc(&d, size * 64), "NewSRS");
            must(zk_bn254_kzg_new_srs_dev(d, size, &alpha, g_srs.g2, nullptr), "NewSRS");
            must(zk_bn254_bases_register_cfg(d, size, 0, 1, -1, &g_srs.handle), "NewSRS");
            (void)zk_dev_free(d);
            lap.lap("export.srs_generate");
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
            lap.lap("export.srs_save");
        }
        g_srs.ready = true;
    }
    *handle = g_srs.handle;
    memcpy(g2, g_srs.g2, sizeof g_srs.g2);
}
// Every export that commits against the SRS passes here after try_load_srs: the second one builds the window tables (see there).
void srs_for_repeat_use() {
    std::lock_guard<std::mutex> lk(g_srs.mu);
    if (g_srs.uses++ < 1 || g_srs.tables || !g_srs.ready) return;
    // Both on the library's background thread (csrc/ctx.hip): THIS call commits against the SRS without the table, as the first one did; calls that start after the
    // table is published use it.  (Round 5 built it here: 17 ms for 1,000,000 points + 39 ms of stream creation on the second call's critical path.)
    must(zk_bn254_bases_build_table_background(g_srs.handle, 0), "SRS window tables");
    g_srs.tables = true;
    (void)zk_warm_session_streams_background();  // a process that proves again also gets the high-priority streams a lean start withheld
}
// plonk.Verify takes two G2 points from the SRS and nothing else (kzg.Verify's pairing check; the reference re-reads the whole file for them,
// backend/plonk/plonk.go:34).  A process that has not proved anything reads them from the file's header on the host: it never starts the HIP runtime.
void srs_g2_for_verify(zk_g2_affine g2[2]) {
    {
        std::lock_guard<std::mutex> lk(g_srs.mu);
        if (!g_srs.ready && !g_srs.g2_only) {
            Lap lap;
            const std::string text = read_srs_text(srs_path());
            if (is_hex_text(text)) {
                lap.lap("export.srs_file_read");
                must(zk_bn254_kzg_srs_g2(text.data(), text.size(), 1, g_srs.g2), "LoadSRS");
                g_srs.g2_only = true;
                lap.lap("export.srs_g2_on_host");
            }
        }
        if (g_srs.ready || g_srs.g2_only) {
            memcpy(g2, g_srs.g2, sizeof g_srs.g2);
            return;
        }
    }
    uint64_t h;  // no usable file: the reference generates and saves one, here on the device
    try_load_srs(&h, g2);
}

char* plonk_prove(GoString acir, View values, const char* pk_hex, size_t pk_len, uint64_t pk_handle, uint64_t srs) {
    char* proof = (char*)malloc(2 * ZK_PLONK_PROOF_BYTES + 1);  // C.CString
    if (!proof) fatal("out of memory");
    must(zk_plonk_prove_with_pk(acir.p, (size_t)acir.n, values.p, values.n, ZK_ACIR_LAYOUT_REFERENCE, pk_hex, pk_len, pk_handle, srs, nullptr, proof), "PlonkProveWithPK");
    proof[2 * ZK_PLONK_PROOF_BYTES] = 0;
    return proof;
}

// The first call of a process: the HIP runtime starts, srs.hex is read and decoded and its window tables are built (0.4 s of driver and GPU work) -- on a
// thread of its own, while this one reads the circuit text (0.2 s of one host core for 2^19 opcodes, no GPU involved): zk_acir_lower_resident leaves the
// lowered circuit where the prover's lookup by content key finds it.  Later calls find both resident.
void load_srs_and_lower(GoString acirJSON, View values, uint64_t* srs, zk_g2_affine g2[2], int with_coefficients = 0) {
    bool first;
    {
        std::lock_guard<std::mutex> lk(g_srs.mu);
        first = !g_srs.ready;
    }
    if (!first) { try_load_srs(srs, g2); srs_for_repeat_use(); return; }
    std::thread loader([&] { try_load_srs(srs, g2); srs_for_repeat_use(); });  // failures end the process (log.Fatal), from whichever thread
    size_t n_values = 0;
    if (zkmi::count_from_hex(values.p, values.n, &n_values))  // errors resurface in the call proper
        (void)zk_acir_lower_resident(acirJSON.p, (size_t)acirJSON.n, n_values, ZK_ACIR_LAYOUT_REFERENCE, with_coefficients);
    loader.join();
}

// ---- Groth16 (backend/groth16/r1cs.go:74-266): no SRS.  A process's first call starts the HIP runtime on a thread of its own while this one reads the RawR1CS text
// (0.25 GB at 2^20 constraints, host only); ProveWithPK's starter goes on to decode the key text (0.37 GB; G1 / G2 decompression on the device) beside it.
std::once_flag g_g16_started;
void groth16_start(GoString rawR1CS, const GoString* pk) {
    // The WHOLE start-up runs inside call_once: a second thread that calls an export while the first is still in start_devices waits here until the flags
    // and the device list are in place (it used to return at once and could initialise libzkmi the default, non-lean way on device 0 before zk_init_devices ran --
    // with ZKMI_DEVICES="2,3" the starter then failed with "entry 0 is already bound").
    std::call_once(g_g16_started, [&] {
        Lap lap;
        std::promise<void> up;
        std::shared_future<void> is_up = up.get_future().share();
        std::thread starter([pk, &up] {
            start_devices(5);  // the five stream slots of a Groth16 proof session (no high-priority streams after a lean start: csrc/ctx.hip)
            up.set_value();
            if (pk) (void)zk_groth16_key_resident(pk->p, (size_t)pk->n);  // (errors resurface in the call proper)
        });
        (void)zk_groth16_lower_resident(rawR1CS.p, (size_t)rawR1CS.n, 0);
        lap.lap("export.raw_lower_beside_start");
        is_up.wait();
        lap.lap("export.hip_start_wait");
        (void)zk_groth16_lower_resident(rawR1CS.p, (size_t)rawR1CS.n, 1);  // the circuit goes to the device while the starter decodes the key
        lap.lap("export.circuit_upload");
        starter.join();
        lap.lap("export.hip_start_and_key_wait");
    });
}

}  // namespace

extern "C" {

char* PlonkProveWithPK(GoString acirJSON, GoString encodedValues, GoString encodedProvingKey) {
    InFlight in_flight;
    uint64_t srs;
    zk_g2_affine g2[2];
    const View values = unquoted(encodedValues);
    load_srs_and_lower(acirJSON, values, &srs, g2);
    return plonk_prove(acirJSON, values, encodedProvingKey.p, (size_t)encodedProvingKey.n, 0, srs);
}

KeyPair PlonkPreprocess(GoString acirJSON, GoString encodedRandomValues) {
    InFlight in_flight;
    uint64_t srs;
    zk_g2_affine g2[2];
    const View values = unquoted(encodedRandomValues);
    load_srs_and_lower(acirJSON, values, &srs, g2, 1);  // a process's first call: the circuit text is lowered (selectors included) while the runtime starts
    size_t pk_len = 0, vk_len = 0;
    must(zk_plonk_preprocess(acirJSON.p, (size_t)acirJSON.n, values.p, values.n, ZK_ACIR_LAYOUT_REFERENCE, srs, nullptr, 0, &pk_len, nullptr, 0, &vk_len, nullptr), "PlonkPreprocess");
    // the key text (0.33 GB at 2^19 gates) is written once, into the C.CString the caller receives
    char* pk = (char*)malloc(pk_len + 1);
    char* vk = (char*)malloc(vk_len + 1);
    if (!pk || !vk) fatal("out of memory");
    must(zk_plonk_preprocess(acirJSON.p, (size_t)acirJSON.n, values.p, values.n, ZK_ACIR_LAYOUT_REFERENCE, srs, pk, pk_len, &pk_len, vk, vk_len, &vk_len, nullptr), "PlonkPreprocess");
    pk[pk_len] = 0;
    vk[vk_len] = 0;
    return KeyPair{pk, vk};
}

char* PlonkProveWithMeta(GoString acirJSON, GoString encodedValues) {
    InFlight in_flight;
    uint64_t srs, h = 0;
    zk_g2_affine g2[2];
    const View values = unquoted(encodedValues);
    load_srs_and_lower(acirJSON, values, &srs, g2, 1);
    size_t pk_len = 0, vk_len = 0;
    must(zk_plonk_preprocess(acirJSON.p, (size_t)acirJSON.n, values.p, values.n, ZK_ACIR_LAYOUT_REFERENCE, srs, nullptr, 0, &pk_len, nullptr, 0, &vk_len, nullptr), "PlonkProveWithMeta");
    std::string pk(pk_len, '\0');
    must(zk_plonk_preprocess(acirJSON.p, (size_t)acirJSON.n, values.p, values.n, ZK_ACIR_LAYOUT_REFERENCE, srs, &pk[0], pk.size(), &pk_len, nullptr, 0, &vk_len, &h), "PlonkProveWithMeta");
    char* proof = plonk_prove(acirJSON, values, nullptr, 0, h, srs);
    (void)zk_bn254_plonk_pk_free(h);
    return proof;
}

unsigned char PlonkVerifyWithMeta(GoString, GoString, GoString) { return 0; }  // main.go:40-42

unsigned char PlonkVerifyWithVK(GoString acirJSON, GoString encodedProof, GoString encodedPublicInputs, GoString encodedVerifyingKey) {
    zk_g2_affine g2[2];
    std::thread srs_reader([&] { srs_g2_for_verify(g2); });  // 64 MB of text (22 ms) beside the circuit's lowering (20 ms at 2^19 opcodes); joined before g2 is used
    struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } join_reader{srs_reader};
    std::vector<uint8_t> proof;
    if (!hex_to_bytes(encodedProof.p, (size_t)encodedProof.n, &proof) || proof.size() != ZK_PLONK_PROOF_BYTES) { fprintf(stderr, "DeserializeProof: not the hex of a PLONK proof\n"); exit(1); }
    // the values arrive indexed by witness (backend.rs:103: get_values_from_witness_tree over all of the circuit's variables); HandleValues keeps the
    // public ones, in witness order (common.go:45-60).  Only those are decoded: the count comes from the vector's header, the positions from the
    // lowering of the circuit (resident after the first call with this text)
    const View vals = unquoted(encodedPublicInputs);
    size_t n_values = 0;
    if (!zkmi::count_from_hex(vals.p, vals.n, &n_values) || (vals.n - 8) / 64 != n_values || (vals.n - 8) % 64) { fprintf(stderr, "DeserializeFelts: invalid felt vector\n"); exit(1); }
    if (!zkmi::all_hex(vals.p + 8, vals.n - 8)) { fprintf(stderr, "DeserializeFelts: invalid felt vector\n"); exit(1); }  // hex.DecodeString sees the whole text
    size_t n_public = 0;
    std::vector<uint32_t> where(16);
    int qrc = zk_acir_public_witnesses(acirJSON.p, (size_t)acirJSON.n, n_values, ZK_ACIR_LAYOUT_REFERENCE, where.data(), where.size(), &n_public);
    if (qrc == ZK_ERR_ARG && n_public > where.size()) {
        where.resize(n_public);
        qrc = zk_acir_public_witnesses(acirJSON.p, (size_t)acirJSON.n, n_values, ZK_ACIR_LAYOUT_REFERENCE, where.data(), where.size(), &n_public);
    }
    must(qrc, "BuildSparseR1CS");
    std::vector<std::vector<uint8_t>> pub_be;
    for (size_t k = 0; k < n_public; k++) {
        std::vector<uint8_t> be;
        if (!hex_to_bytes(vals.p + 8 + 64 * (size_t)where[k], 64, &be)) { fprintf(stderr, "DeserializeFelts: invalid felt vector\n"); exit(1); }
        pub_be.push_back(be);
    }
    std::vector<zk_fr> pub;
    if (!be_to_mont(pub_be, &pub)) { fprintf(stderr, "DeserializeFelts: invalid fr.Element encoding\n"); exit(1); }
    int ok = 0;
    srs_reader.join();
    const int rc = zk_bn254_plonk_verify(proof.data(), encodedVerifyingKey.p, (size_t)encodedVerifyingKey.n, 1, g2, pub.data(), pub.size(), &ok);
    if (rc == ZK_ERR_LEN) return 0;  // plonk.Verify's "invalid witness size" is an error value upstream, i.e. `false` (plonk.go:47-50)
    must(rc, "PlonkVerifyWithVK");
    return ok ? 1 : 0;
}

// ---- Groth16 (backend/groth16/r1cs.go:74-266)
char* ProveWithMeta(GoString rawR1CS) {
    InFlight in_flight;
    groth16_start(rawR1CS, nullptr);
    std::string proof(256, '\0');
    must(zk_groth16_prove_with_meta(rawR1CS.p, (size_t)rawR1CS.n, nullptr, nullptr, &proof[0]), "ProveWithMeta");
    return c_string(proof);
}
char* ProveWithPK(GoString rawR1CS, GoString encodedProvingKey) {
    InFlight in_flight;
    groth16_start(rawR1CS, &encodedProvingKey);
    std::string proof(256, '\0');
    must(zk_groth16_prove_with_pk(rawR1CS.p, (size_t)rawR1CS.n, encodedProvingKey.p, (size_t)encodedProvingKey.n, 0, nullptr, &proof[0]), "ProveWithPK");
    return c_string(proof);
}
KeyPair Preprocess(GoString rawR1CS) {
    InFlight in_flight;
    groth16_start(rawR1CS, nullptr);
    size_t pk_len = 0, vk_len = 0;
    must(zk_groth16_preprocess(rawR1CS.p, (size_t)rawR1CS.n, nullptr, nullptr, 0, &pk_len, nullptr, 0, &vk_len, nullptr), "Preprocess");  // Setup runs here; the key waits
    // the key text (0.37 GB at 2^20 constraints) is written once, into the C.CString the caller receives
    char* pk = (char*)malloc(pk_len + 1);
    char* vk = (char*)malloc(vk_len + 1);
    if (!pk || !vk) fatal("out of memory");
    must(zk_groth16_preprocess(rawR1CS.p, (size_t)rawR1CS.n, nullptr, pk, pk_len, &pk_len, vk, vk_len, &vk_len, nullptr), "Preprocess");
    pk[pk_len] = 0;
    vk[vk_len] = 0;
    return KeyPair{pk, vk};
}
// upstream's sketch runs a fresh Setup and verifies against ITS key (r1cs.go:145-174): no proof made elsewhere can pass -- `false`, like PlonkVerifyWithMeta
unsigned char VerifyWithMeta(GoString, GoString) { return 0; }
unsigned char VerifyWithVK(GoString rawR1CS, GoString encodedProof, GoString encodedVerifyingKey) {
    std::vector<uint8_t> proof;
    if (!hex_to_bytes(encodedProof.p, (size_t)encodedProof.n, &proof) || proof.size() != 128) { fprintf(stderr, "DeserializeProof: not the hex of a Groth16 proof\n"); exit(1); }
    // buildWitnesses' public part: the values of the public wires -- read on the host against the circuit's (resident) lowering; a process that only
    // verifies never starts the HIP runtime
    size_t n_public = 0;
    std::vector<zk_fr> pub(16);
    int rc = zk_groth16_public_inputs(rawR1CS.p, (size_t)rawR1CS.n, pub.data(), pub.size(), &n_public);
    if (rc == ZK_ERR_ARG && n_public > pub.size()) {
        pub.resize(n_public);
        rc = zk_groth16_public_inputs(rawR1CS.p, (size_t)rawR1CS.n, pub.data(), pub.size(), &n_public);
    }
    must(rc, "buildR1CS");
    int ok = 0;
    rc = zk_bn254_groth16_verify(proof.data(), encodedVerifyingKey.p, (size_t)encodedVerifyingKey.n, 1, pub.data(), n_public, &ok);
    if (rc == ZK_ERR_LEN) return 0;
    must(rc, "VerifyWithVK");
    return ok ? 1 : 0;
}

}  // extern "C"
