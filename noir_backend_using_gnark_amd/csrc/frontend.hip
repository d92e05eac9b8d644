// The callers either side of the hot path (SURVEY §8 rows f3 and the outer boundary of §8b): what the reference's Go shim does between the
// strings Noir hands over and gnark's prover -- restated on top of the device path, so that the reference's exported entry points have
// semantic equivalents here that never leave the GPU between the witness text and the proof bytes:
//     PlonkPreprocess(acirJSON, encodedValues)              -> (hex pk, hex vk)    /root/reference/gnark_backend_ffi/main.go:58-78
//     PlonkProveWithPK(acirJSON, encodedValues, encodedPK)  -> hex proof           /root/reference/gnark_backend_ffi/main.go:24-37
// through
//     acir.ACIR JSON                                  gnark_backend_ffi/acir/acir.go:17-75, opcode/arithmetic_opcode.go:18-83, term/*.go
//     BuildSparseR1CS / handleArithmeticOpcode        backend/plonk/sparse_r1cs.go:18-107 (one gate per arithmetic opcode: MulTerms[0] only;
//                                                     SimpleTerms of length 1 -> qO, 2 -> qL qR, 3 -> qL qR qO; directives / black boxes emit nothing)
//     HandleValues / BuildWitnesses                   backend/common.go:22-76 (public variables first, in witness order, then the secret ones)
//     DeserializeFelts                                internal/backend/helpers.go:24-33 (wire.hip, on the device)
//     Serialize / DeserializeProvingKey, VerifyingKey internal/backend/helpers.go:49-94 (plonk.hip / keyio.hip, on the device)
// Differences, on purpose: the SRS is a handle the caller keeps resident (the reference re-reads srs.hex on every call, plonk.go:16,34,58);
// failures are error codes (the reference log.Fatal()s); the nine blinding scalars can be pinned (NULL: drawn from the OS generator like
// upstream's fr.SetRandom).  Variable layout: ZK_ACIR_LAYOUT_REFERENCE (the default, what libgnark_backend.so uses) reproduces HandleValues
// literally -- with two or more public inputs it appends one secret variable per (witness, non-matching public input) and the gates use the
// LAST copy (common.go:59-68), so keys and proofs are interchangeable with the reference's; ZK_ACIR_LAYOUT_ONE_VAR_PER_WITNESS is the layout
// without the duplicates (the two coincide for zero or one public input).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <future>
#include <list>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "acir_host.hpp"  // the text side (host only, sanitizer- and mutation-tested on the CPU: tests/cpp/parser_fuzz.cpp)
#include <atomic>
#include "text_host.hpp"
#include "ctx.hpp"
#include "ff.hpp"
#include "host_ff.hpp"

namespace zkmi {

static int lower(const char* json, size_t len, size_t n_values, int layout, bool with_coeffs, Gates* G) {
    std::string err;
    const int rc = lower_acir(json, len, n_values, layout, with_coeffs, G, &err);
    return rc == ZK_OK ? ZK_OK : set_err(rc, "%s", err.c_str());
}

__global__ void k_gather_fr(const Fr* __restrict__ vals, const uint32_t* __restrict__ order, size_t n, Fr* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = vals[order[i]];
}

static int count_of(const char* hex, size_t len, size_t* n) {
    if (len < 8) return set_err(ZK_ERR_ARG, "felt vector: %zu characters cannot hold the 4-byte count", len);
    if (!count_from_hex(hex, len, n)) return set_err(ZK_ERR_ARG, "felt vector: invalid hex character in the count");
    return ZK_OK;
}

static int circuit_of(const Gates& G, zk_plonk_circuit* c) {
    memset(c, 0, sizeof *c);
    c->n_public = G.n_public;
    c->n_constraints = G.xa.size();
    c->n_vars = G.n_vars;
    c->ql = G.ql.data(); c->qr = G.qr.data(); c->qo = G.qo.data(); c->qm = G.qm.data(); c->qk = G.qk.data();
    c->xa = G.xa.data(); c->xb = G.xb.data(); c->xc = G.xc.data();
    return ZK_OK;
}

// wall-clock sections of the export path, reported beside the kernels when profiling is on (zk_profile_*; bench.py's `export_path` block)
int felts_decode_hex_on_slot(Slot* s, hipStream_t st, const void* d_text, size_t text_len, void* d_out, size_t cap, size_t n, int to_mont, int* d_status);  // wire.hip
struct InFlight {  // background jobs wait for calls in flight (ctx.hip)
    InFlight() { zk_background_hold(1); }
    ~InFlight() { zk_background_hold(-1); }
    InFlight(const InFlight&) = delete;
    InFlight& operator=(const InFlight&) = delete;
};
struct Phase {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char* name) {
        const auto t1 = std::chrono::steady_clock::now();
        prof_host(name, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// ------------------------------------------------------------------------------------------------ what stays resident between calls
// The reference re-reads everything on every call: the ACIR text is unmarshalled and lowered, the proving key is hex-decoded and ReadFrom'd
// (main.go:24-37 -> backend/plonk/plonk.go:53-73), the SRS file is re-read (plonk.go:16,34,58).  At 2^19 gates that is 0.24 GB of JSON and 0.33 GB of
// key text in front of a 10 ms prover.  Here both texts are identified by a 128-bit content key (acir_host.hpp: sixteen threads read the 0.57 GB in a few
// ms) and what was derived from them stays resident:
//   Lowered   the circuit's wiring (xa, xb, xc), HandleValues' variable order in HBM, and the per-proof buffers of the witness vector
//   CachedKey the decoded proving key with its nine big-coset forms ("device-resident bases cache keyed by pk hash")
// both least-recently-used within ZKMI_TABLE_CAP_GB (the bound the window tables of resident keys already obey; 0 = keep nothing between calls).
// A second call with the same texts costs: two content keys, the witness vector (33 MB of hex, decoded on the device), the prover.
struct Lowered {
    ContentKey key;
    size_t n_values = 0, text_len = 0;  // text_len: characters of the ACIR text this was lowered from
    int layout = 0, entry = 0;  // entry: the device entry whose HBM holds d_order / d_vals / d_sol
    size_t n_public = 0, n_vars = 0, n_gates = 0;
    std::vector<uint32_t> xa, xb, xc, order_public;  // host copies: a key that is not resident yet is read against the wiring; the verifier's public-witness list
    std::vector<uint32_t> order;                     // kept until it is uploaded
    std::mutex work;                                 // one proof at a time per circuit (the key serialises them anyway)
    uint32_t* d_order = nullptr;
    void *d_vals = nullptr, *d_sol = nullptr;
    size_t bytes() const { return (xa.size() * 3 + order_public.size() + order.size()) * 4 + (d_order ? n_vars * 36 + n_values * 32 : 0); }
    ~Lowered() {
        for (void* q : {(void*)d_order, d_vals, d_sol})
            if (q) (void)hipFree(q);
    }
    int device_buffers() {  // under `work`
        if (d_order) return ZK_OK;
        ZK_TRY(ensure_init());
        ZK_HIP(hipMalloc((void**)&d_order, (n_vars ? n_vars : 1) * 4));
        ZK_HIP(hipMalloc(&d_vals, (n_values ? n_values : 1) * 32));
        ZK_HIP(hipMalloc(&d_sol, (n_vars ? n_vars : 1) * 32));
        if (n_vars) ZK_HIP(hipMemcpy(d_order, order.data(), n_vars * 4, hipMemcpyHostToDevice));
        std::vector<uint32_t>().swap(order);
        return ZK_OK;
    }
};
struct CachedKey {
    ContentKey pk, acir;
    size_t n_values = 0;
    int layout = 0;
    uint64_t srs = 0, handle = 0;
    size_t bytes = 0, text_len = 0;  // text_len: characters of the key text this entry was read from
    int in_use = 0;
    unsigned proofs = 0;  // proofs made with this resident key (zk_plonk_prove_with_pk): the 16th gives it the SRS's Lagrange form
};
static std::mutex g_cache_mu;
static constexpr unsigned kLagrangeAfterProofs = 16;
static std::list<std::shared_ptr<Lowered>> g_lowered;  // most recently used first
static std::list<CachedKey> g_keys;
static size_t cache_cap_bytes() {
    static const size_t cap_gb = (size_t)zk_env_bounded("ZKMI_TABLE_CAP_GB", 128, 0, 1024);
    return cap_gb << 30;
}
// drops least-recently-used entries until `extra` more bytes fit (entries in use stay); keys first (1 GB each at 2^19 gates), then circuits
static size_t g16_cache_bytes_locked();  // the Groth16 entries (further down): both caches count against the ONE bound
static size_t plonk_cache_bytes_locked();
static void cache_trim_locked(size_t extra, std::vector<uint64_t>* to_free) {
    const size_t cap = cache_cap_bytes();
    auto total = [&]() {
        size_t t = extra + g16_cache_bytes_locked();
        for (auto& k : g_keys) t += k.bytes;
        for (auto& l : g_lowered) t += l->bytes();
        return t;
    };
    for (auto it = g_keys.end(); it != g_keys.begin() && (total() > cap || g_keys.size() >= 8);) {
        --it;
        if (it->in_use) continue;
        to_free->push_back(it->handle);
        it = g_keys.erase(it);
    }
    while (!g_lowered.empty() && (total() > cap || g_lowered.size() >= 8)) {
        if (g_lowered.back().use_count() > 1) break;  // a call is using it
        g_lowered.pop_back();
    }
}
static void free_handles(const std::vector<uint64_t>& hs) {
    for (uint64_t h : hs) (void)zk_bn254_plonk_pk_free(h);
}

// One fully lowered circuit (coefficients included) waits here between the two calls of a preprocess -- the size query and the call that writes the
// key -- so that 0.24 GB of JSON is read once, not twice.  Taken by the first request for it, replaced by the next size query.
struct Stash {
    ContentKey key;
    size_t n_values = 0;
    int layout = 0;
    std::unique_ptr<Gates> gates;
};
static Stash g_stash;
static void stash_put(const Lowered& L, Gates&& G) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    g_stash.key = L.key;
    g_stash.n_values = L.n_values;
    g_stash.layout = L.layout;
    g_stash.gates.reset(new Gates(std::move(G)));
}

// the circuit behind an ACIR text: from the cache, or lowered now.  full != NULL: the caller also wants the gates with their coefficients (Setup).
static int lowered_get(const char* acir_json, size_t acir_len, size_t n_values, int layout, std::shared_ptr<Lowered>* out, Gates* full, const ContentKey* known = nullptr) {
    Phase ph;
    const ContentKey key = known ? *known : content_key(acir_json, acir_len);
    if (!known) ph.lap("export.acir_content_key");
    bool have_full = false;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        if (full && g_stash.gates && g_stash.key == key && g_stash.n_values == n_values && g_stash.layout == layout) {
            *full = std::move(*g_stash.gates);
            g_stash.gates.reset();
            have_full = true;
        }
        if (!full || have_full)
            for (auto it = g_lowered.begin(); it != g_lowered.end(); ++it)
                if ((*it)->key == key && (*it)->n_values == n_values && (*it)->layout == layout && (*it)->entry == current_entry()) {
                    g_lowered.splice(g_lowered.begin(), g_lowered, it);
                    *out = g_lowered.front();
                    return ZK_OK;
                }
    }
    Gates local;
    Gates* G = full ? full : &local;
    if (!have_full) {
        ZK_TRY(lower(acir_json, acir_len, n_values, layout, full != nullptr, G));
        ph.lap(full ? "export.acir_parse_lower_with_coefficients" : "export.acir_parse_lower");
    }
    auto L = std::make_shared<Lowered>();
    L->key = key;
    L->n_values = n_values;
    L->text_len = acir_len;
    L->layout = layout;
    L->entry = current_entry();
    L->n_public = G->n_public;
    L->n_vars = G->n_vars;
    L->n_gates = G->xa.size();
    L->order_public.assign(G->order.begin(), G->order.begin() + G->n_public);
    if (full) { L->xa = G->xa; L->xb = G->xb; L->xc = G->xc; L->order = G->order; }
    else { L->xa.swap(G->xa); L->xb.swap(G->xb); L->xc.swap(G->xc); L->order.swap(G->order); }
    std::vector<uint64_t> dead;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (auto it = g_lowered.begin(); it != g_lowered.end(); ++it)
            if ((*it)->key == key && (*it)->n_values == n_values && (*it)->layout == layout && (*it)->entry == current_entry()) { g_lowered.erase(it); break; }  // re-lowered with coefficients
        if (cache_cap_bytes()) {
            cache_trim_locked(L->bytes() + L->n_vars * 36 + L->n_values * 32, &dead);
            g_lowered.push_front(L);
        }
    }
    free_handles(dead);
    *out = L;
    return ZK_OK;
}

// the resident key behind a key text: from the cache, or decoded now against the circuit's wiring.  *entry_handle stays valid until key_release.
static int key_get(const char* pk_hex, size_t pk_len, const Lowered& L, uint64_t srs, uint64_t* handle, bool* cached, const ContentKey* known = nullptr) {
    Phase ph;
    const ContentKey key = known ? *known : content_key(pk_hex, pk_len);  // (known: computed beside the circuit text's key)
    ph.lap("export.pk_content_key");
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (auto it = g_keys.begin(); it != g_keys.end(); ++it)
            if (it->pk == key && it->acir == L.key && it->n_values == L.n_values && it->layout == L.layout && it->srs == srs) {
                it->in_use++;
                *handle = it->handle;
                *cached = true;
                g_keys.splice(g_keys.begin(), g_keys, it);
                return ZK_OK;
            }
    }
    *cached = false;
    uint64_t h = 0;
    int rc = zk_bn254_plonk_pk_read(pk_hex, pk_len, 1, L.n_vars, L.n_gates, L.xa.data(), L.xb.data(), L.xc.data(), srs, &h);
    if (rc == ZK_ERR_HIP) {  // out of HBM with idle keys resident: let them go and try once more
        std::vector<uint64_t> dead;
        {
            std::lock_guard<std::mutex> lk(g_cache_mu);
            for (auto it = g_keys.begin(); it != g_keys.end();)
                if (!it->in_use) { dead.push_back(it->handle); it = g_keys.erase(it); } else ++it;
        }
        if (!dead.empty()) {
            free_handles(dead);
            rc = zk_bn254_plonk_pk_read(pk_hex, pk_len, 1, L.n_vars, L.n_gates, L.xa.data(), L.xb.data(), L.xc.data(), srs, &h);
        }
    }
    ZK_TRY(rc);
    *handle = h;
    size_t bytes = 0;
    (void)zk_bn254_plonk_pk_bytes(h, &bytes);
    if (!cache_cap_bytes() || bytes > cache_cap_bytes()) return ZK_OK;  // not kept: the caller frees it (*cached stays false)
    std::vector<uint64_t> dead;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        cache_trim_locked(bytes, &dead);
        CachedKey e;
        e.pk = key; e.acir = L.key; e.n_values = L.n_values; e.layout = L.layout; e.srs = srs; e.handle = h; e.bytes = bytes; e.in_use = 1; e.text_len = pk_len;
        g_keys.push_front(e);
        *cached = true;
    }
    free_handles(dead);
    return ZK_OK;
}
static void key_release(uint64_t handle) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    for (auto& k : g_keys)
        if (k.handle == handle) { if (k.in_use > 0) k.in_use--; return; }
}
// a key Setup has just produced enters the cache under the content key of the text it was written as: the PlonkProveWithPK that follows finds it
static void key_adopt(const char* pk_hex, size_t pk_len, const Lowered& L, uint64_t srs, uint64_t h, bool* adopted) {
    *adopted = false;
    size_t bytes = 0;
    if (zk_bn254_plonk_pk_bytes(h, &bytes) != ZK_OK || !cache_cap_bytes() || bytes > cache_cap_bytes()) return;
    const ContentKey key = content_key(pk_hex, pk_len);
    std::vector<uint64_t> dead;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (auto& k : g_keys)
            if (k.pk == key && k.acir == L.key && k.n_values == L.n_values && k.layout == L.layout && k.srs == srs) return;  // already resident
        cache_trim_locked(bytes, &dead);
        CachedKey e;
        e.pk = key; e.acir = L.key; e.n_values = L.n_values; e.layout = L.layout; e.srs = srs; e.handle = h; e.bytes = bytes; e.in_use = 0; e.text_len = pk_len;
        g_keys.push_front(e);
        *adopted = true;
    }
    free_handles(dead);
}

// One proof on a resident circuit with a resident key (under L.work): DeserializeFelts on the device, the public-first gather (BuildWitnesses), the prover.
static int plonk_prove_on(Lowered& L, const char* values_hex, size_t values_len, size_t n_values, uint64_t h, const zk_fr* blinders, uint8_t* proof) {
    Phase ph;
    ZK_TRY(L.device_buffers());
    ph.lap("export.circuit_to_device");
    size_t n_dec = 0;
    ZK_TRY(zk_bn254_felts_decode_hex(values_hex, values_len, L.d_vals, n_values ? n_values : 1, &n_dec));
    ph.lap("export.values_decode");
    {
        SlotGuard g;
        ZK_TRY(acquire_slot(&g.s));
        hipStream_t st = g.s->stream;
        if (L.n_vars)
            ZK_LAUNCH(g.s, st, "witness_gather", k_gather_fr, dim3((unsigned)((L.n_vars + 255) / 256)), dim3(256), 0, (const Fr*)L.d_vals, (const uint32_t*)L.d_order, L.n_vars, (Fr*)L.d_sol);
        ZK_TRY(slot_sync(g.s, st));
    }
    ph.lap("export.witness_gather");
    {   // the key must be the key of THIS circuit: same public / variable / gate counts (a key of another shape would read the solution out of bounds)
        size_t kp = 0, kv = 0, kc = 0;
        ZK_TRY(zk_bn254_plonk_pk_info(h, nullptr, &kp, &kc, &kv));
        if (kp != L.n_public || kv != L.n_vars || kc != L.n_gates)
            return set_err(ZK_ERR_ARG, "proving key is for %zu public / %zu variables / %zu gates, the circuit has %zu / %zu / %zu", kp, kv, kc, L.n_public, L.n_vars, L.n_gates);
    }
    ZK_TRY(zk_bn254_plonk_prove(h, L.d_sol, L.n_vars, 1, blinders, nullptr, proof));
    ph.lap("export.plonk_prove");
    return ZK_OK;
}
static void plonk_proof_hex(const uint8_t* proof, char* out) {
    static const char dig[] = "0123456789abcdef";
    for (size_t i = 0; i < ZK_PLONK_PROOF_BYTES; i++) {
        out[2 * i] = dig[proof[i] >> 4];
        out[2 * i + 1] = dig[proof[i] & 15];
    }
}

}  // namespace zkmi

using namespace zkmi;

extern "C" {

// PlonkPreprocess (main.go:58-78): ACIR + the witness-value vector (only its length and the public/secret split matter to Setup) -> hex of
// ProvingKey.WriteTo and hex of VerifyingKey.WriteTo (= the first 368 bytes of the former).  pk_hex_out may be NULL with pk_cap = 0 to query
// the sizes.  The key also stays resident: *pk_handle (optional) can be handed to zk_bn254_plonk_prove directly; without it the key enters the
// export cache under the content key of pk_hex_out, so that the zk_plonk_prove_with_pk which follows in this process neither decodes nor rebuilds it.
int zk_plonk_preprocess(const char* acir_json, size_t acir_len, const char* values_hex, size_t values_len, int layout, uint64_t srs_handle, char* pk_hex_out,
                        size_t pk_cap, size_t* pk_len, char* vk_hex_out, size_t vk_cap, size_t* vk_len, uint64_t* pk_handle) {
    InFlight _in_flight;
    if (!acir_json || !values_hex || !pk_len || !vk_len) return set_err(ZK_ERR_ARG, "null pointer");
    ZK_ON_ENTRY_OF(srs_handle);
    size_t n_values = 0;
    ZK_TRY(count_of(values_hex, values_len, &n_values));
    if (values_len != 8 + 64 * n_values) return set_err(ZK_ERR_LEN, "felt vector: %zu characters, the count says %zu felts", values_len, n_values);
    std::shared_ptr<Lowered> L;
    Gates G;
    ZK_TRY(lowered_get(acir_json, acir_len, n_values, layout, &L, &G));
    // sizes first (a call with pk_hex_out == NULL only asks for them): domain n = next power of two >= gates + public inputs
    size_t n = 1;
    while (n < L->n_gates + L->n_public) n <<= 1;
    *pk_len = 2 * (704 + 9 * (4 + 32 * n) + 24 * n);
    *vk_len = 2 * 368;
    if (!pk_hex_out) { stash_put(*L, std::move(G)); return ZK_OK; }  // the lowered circuit waits for the call that writes the key
    if (pk_cap < *pk_len || (vk_hex_out && vk_cap < *vk_len)) return set_err(ZK_ERR_ARG, "outputs hold %zu / %zu characters, %zu / %zu needed", pk_cap, vk_cap, *pk_len, *vk_len);
    Phase ph;
    zk_plonk_circuit c;
    circuit_of(G, &c);
    uint64_t h = 0;
    ZK_TRY(zk_bn254_plonk_setup(&c, srs_handle, &h, nullptr));
    ph.lap("export.plonk_setup");
    int rc = zk_bn254_plonk_pk_write(h, 1, pk_hex_out, pk_cap, pk_len);
    ph.lap("export.pk_write_hex");
    if (rc == ZK_OK && vk_hex_out) memcpy(vk_hex_out, pk_hex_out, *vk_len);  // ProvingKey.WriteTo starts with VerifyingKey.WriteTo
    if (pk_handle && rc == ZK_OK) { *pk_handle = h; return ZK_OK; }
    bool adopted = false;
    if (rc == ZK_OK) key_adopt(pk_hex_out, *pk_len, *L, srs_handle, h, &adopted);
    ph.lap("export.pk_adopt");
    if (!adopted) (void)zk_bn254_plonk_pk_free(h);
    return rc;
}

// PlonkProveWithPK (main.go:24-37): ACIR + hex witness values + hex proving key -> hex of Proof.WriteTo (2 * 548 characters, no terminator).
// pk_hex may be NULL when pk_handle names a resident key (zk_plonk_preprocess / zk_bn254_plonk_pk_read) -- the reference deserialises the
// key on every call; here a key text seen before (same circuit, same SRS) is found resident by its content key.
// blinders: 9 scalars or NULL (drawn from /dev/urandom, as upstream draws them with fr.SetRandom).
int zk_plonk_prove_with_pk(const char* acir_json, size_t acir_len, const char* values_hex, size_t values_len, int layout, const char* pk_hex, size_t pk_len,
                           uint64_t pk_handle, uint64_t srs_handle, const zk_fr* blinders, char proof_hex_out[2 * ZK_PLONK_PROOF_BYTES]) {
    InFlight _in_flight;
    if (!acir_json || !values_hex || !proof_hex_out || (!pk_hex && !pk_handle)) return set_err(ZK_ERR_ARG, "null pointer");
    ZK_ON_ENTRY_OF(pk_handle ? pk_handle : srs_handle);
    size_t n_values = 0;
    ZK_TRY(count_of(values_hex, values_len, &n_values));
    // the two texts are identified by content (190 MB + 327 MB at 2^19 gates: 1.4 + 1.7 ms of sixteen threads each): the key text's key beside the circuit's
    std::future<ContentKey> pk_key;
    if (pk_hex) pk_key = std::async(std::launch::async, [pk_hex, pk_len] { return content_key(pk_hex, pk_len); });
    zk_fr rnd[9];
    if (!blinders) {
        FILE* f = fopen("/dev/urandom", "rb");
        uint8_t raw[9 * 32];
        if (!f || fread(raw, 1, sizeof raw, f) != sizeof raw) {
            if (f) fclose(f);
            return set_err(ZK_ERR_ARG, "no randomness source for the blinding scalars");
        }
        fclose(f);
        for (int i = 0; i < 9; i++) {
            uint64_t t[4];
            memcpy(t, raw + 32 * i, 32);
            t[3] &= 0x3fffffffffffffffULL;  // < 2^254, then one conditional subtraction: uniform enough for blinding (upstream: rejection sampling)
            while (HFr::geq_mod(t)) HFr::sub_mod(t);
            HFr m = HFr{{t[0], t[1], t[2], t[3]}}.to_mont();
            memcpy(&rnd[i], &m, 32);
        }
        blinders = rnd;
    }
    uint8_t proof[ZK_PLONK_PROOF_BYTES];
    ContentKey acir_ck, pk_ck;
    bool have_acir_ck = false, have_pk_ck = false;
    // A warm call: a resident circuit was lowered from a text of this length and a resident key of it was read from a text of this length.  The proof is made
    // with those while the 0.5 GB of text are still being compared (3 ms in front of a 15 ms prover at 2^19 gates otherwise) and leaves this call only if BOTH
    // content keys say "same"; anything else -- and the proof that would trigger the key's Lagrange form -- takes the ordinary path below.
    static const bool speculate = ZK_EXP("ZKMI_EXPORT_SPECULATE", 1) != 0;
    if (speculate && pk_hex) {
        std::shared_ptr<Lowered> cand;
        uint64_t spec = 0;
        ContentKey spec_ck;
        {
            std::lock_guard<std::mutex> lk(g_cache_mu);
            for (auto& l : g_lowered)
                if (l->text_len == acir_len && l->n_values == n_values && l->layout == layout && l->entry == current_entry()) { cand = l; break; }
            if (cand)
                for (auto& k : g_keys)
                    if (k.text_len == pk_len && k.acir == cand->key && k.n_values == n_values && k.layout == layout && k.srs == srs_handle && k.proofs >= 1 &&
                        k.proofs + 1 != kLagrangeAfterProofs) {
                        k.in_use++;
                        spec = k.handle;
                        spec_ck = k.pk;
                        break;
                    }
        }
        if (spec) {
            std::future<ContentKey> acir_key = std::async(std::launch::async, [acir_json, acir_len] {
                Phase ph;
                const ContentKey k = content_key(acir_json, acir_len);
                ph.lap("export.acir_content_key");
                return k;
            });
            int rp = ensure_init();
            if (rp == ZK_OK) {
                std::lock_guard<std::mutex> work(cand->work);
                rp = plonk_prove_on(*cand, values_hex, values_len, n_values, spec, blinders, proof);
            }
            acir_ck = acir_key.get();
            pk_ck = pk_key.get();
            have_acir_ck = have_pk_ck = true;
            const bool ours = rp == ZK_OK && acir_ck == cand->key && pk_ck == spec_ck;
            {
                std::lock_guard<std::mutex> lk(g_cache_mu);
                for (auto it = g_keys.begin(); it != g_keys.end(); ++it)
                    if (it->handle == spec) {
                        it->in_use--;
                        if (ours) { it->proofs++; g_keys.splice(g_keys.begin(), g_keys, it); }
                        break;
                    }
                if (ours)
                    for (auto it = g_lowered.begin(); it != g_lowered.end(); ++it)
                        if (it->get() == cand.get()) { g_lowered.splice(g_lowered.begin(), g_lowered, it); break; }
            }
            if (ours) {
                plonk_proof_hex(proof, proof_hex_out);
                return ZK_OK;
            }
        }
    }
    std::shared_ptr<Lowered> L;
    ZK_TRY(lowered_get(acir_json, acir_len, n_values, layout, &L, nullptr, have_acir_ck ? &acir_ck : nullptr));
    ZK_TRY(ensure_init());
    std::lock_guard<std::mutex> work(L->work);
    Phase ph;
    uint64_t h = pk_handle;
    bool cached = false;
    if (pk_hex) {
        if (!have_pk_ck) pk_ck = pk_key.get();
        ZK_TRY(key_get(pk_hex, pk_len, *L, srs_handle, &h, &cached, &pk_ck));
    }
    struct Done {  // whatever happens below: a cached key is unpinned, an uncached one freed
        uint64_t h; bool from_text, cached;
        ~Done() { if (from_text) { if (cached) key_release(h); else (void)zk_bn254_plonk_pk_free(h); } }
    } done{h, pk_hex != nullptr, cached};
    ph.lap("export.pk_resident");
    if (cached) {
        // A key that keeps being used is worth the one-time Lagrange form of its SRS (lagrange.hip: 0.15 s at 2^19 gates, then l, r, o are committed from the
        // witness values -- 1.9 of 15.6 ms there): built when the 16th proof with it is asked for.  A failure (SRS spread over several GPUs) leaves the key as it was.
        bool build = false;
        {
            std::lock_guard<std::mutex> lk(g_cache_mu);
            for (auto& k : g_keys)
                if (k.handle == h) { build = ++k.proofs == kLagrangeAfterProofs; break; }
        }
        if (build) {
            if (zk_bn254_plonk_pk_lagrange_srs(h) == ZK_OK) {
                size_t bytes = 0;
                (void)zk_bn254_plonk_pk_bytes(h, &bytes);
                std::lock_guard<std::mutex> lk(g_cache_mu);
                for (auto& k : g_keys)
                    if (k.handle == h) k.bytes = bytes;
            }
            ph.lap("export.pk_lagrange_srs");
        }
    }
    ZK_TRY(plonk_prove_on(*L, values_hex, values_len, n_values, h, blinders, proof));
    plonk_proof_hex(proof, proof_hex_out);
    return ZK_OK;
}

// Lowers a circuit text into the resident cache without touching a device (a caller that has other start-up work in flight -- the shim's SRS load -- runs this
// beside it): the zk_plonk_prove_with_pk / zk_acir_public_witnesses that follows finds it by content key.  The lowering is keyed by device entry too: this is for
// the entry the calling thread is on.
int zk_acir_lower_resident(const char* acir_json, size_t acir_len, size_t n_values, int layout, int with_coefficients) {
    if (!acir_json) return set_err(ZK_ERR_ARG, "null pointer");
    std::shared_ptr<Lowered> L;
    if (!with_coefficients) return lowered_get(acir_json, acir_len, n_values, layout, &L, nullptr);
    Gates G;  // the selectors too: they wait in the stash for zk_plonk_preprocess's size query, as that query's own lowering does for the call that writes the key
    ZK_TRY(lowered_get(acir_json, acir_len, n_values, layout, &L, &G));
    stash_put(*L, std::move(G));
    return ZK_OK;
}

// The public inputs of a circuit as the verifier needs them: out[k] = 0-based index, into the witness-value vector, of public variable k (HandleValues'
// first loop, common.go:45-60).  *n_public comes back even when cap is too small (ZK_ERR_ARG then).  Uses the resident lowering of the text when there is one.
int zk_acir_public_witnesses(const char* acir_json, size_t acir_len, size_t n_values, int layout, uint32_t* out, size_t cap, size_t* n_public) {
    if (!acir_json || !n_public) return set_err(ZK_ERR_ARG, "null pointer");
    std::shared_ptr<Lowered> L;
    ZK_TRY(lowered_get(acir_json, acir_len, n_values, layout, &L, nullptr));
    *n_public = L->n_public;
    if (L->n_public > cap || (L->n_public && !out)) return set_err(ZK_ERR_ARG, "%zu public inputs, the output holds %zu", L->n_public, cap);
    if (L->n_public) memcpy(out, L->order_public.data(), L->n_public * 4);
    return ZK_OK;
}

// buildR1CS of the reference's intended Groth16 FFI (backend/groth16/r1cs.go:9-72, commented out there; payload RawR1CS of
// src/gnark_backend_wrapper/groth16/acir_to_r1cs.rs:18-60) -- the text side is acir_host.hpp's raw_r1cs_build.  Out: a resident R1CS (zk_bn254_r1cs_*)
// and the full wire vector in HBM (*d_witness: n_wires Montgomery elements, to be released with zk_dev_free) -- ready for zk_bn254_groth16_setup /
// _prove_r1cs(on_device = 1).
static int r1cs_of_built(const RawR1CSBuilt& B, uint64_t* handle) {
    zk_r1cs r;
    memset(&r, 0, sizeof r);
    r.n_constraints = B.ptr[0].size() - 1;
    r.n_wires = 1 + B.n_values + B.prod_a.size();
    r.n_public = B.n_public;
    r.l_ptr = B.ptr[0].data(); r.l_idx = B.idx[0].data(); r.l_val = (const zk_fr*)B.val[0].data();
    r.r_ptr = B.ptr[1].data(); r.r_idx = B.idx[1].data(); r.r_val = (const zk_fr*)B.val[1].data();
    r.o_ptr = B.ptr[2].data(); r.o_idx = B.idx[2].data(); r.o_val = (const zk_fr*)B.val[2].data();
    return zk_bn254_r1cs_load(&r, handle);
}
int zk_groth16_r1cs_from_raw(const char* raw_json, size_t len, uint64_t* r1cs_handle, void** d_witness, size_t* n_wires, size_t* n_public) {
    if (!raw_json || !r1cs_handle || !d_witness) return set_err(ZK_ERR_ARG, "null pointer");
    RawR1CSBuilt B;
    {
        std::string err;
        const int rc = raw_r1cs_build(raw_json, len, &B, &err);
        if (rc != ZK_OK) return set_err(rc, "%s", err.c_str());
    }
    ZK_TRY(r1cs_of_built(B, r1cs_handle));
    void* d = nullptr;
    int rc = zk_dev_alloc(&d, B.wires.size() * 32);
    if (rc == ZK_OK) rc = zk_dev_h2d(d, B.wires.data(), B.wires.size() * 32);
    if (rc != ZK_OK) {
        if (d) (void)zk_dev_free(d);
        (void)zk_bn254_r1cs_free(*r1cs_handle);
        return rc;
    }
    *d_witness = d;
    if (n_wires) *n_wires = B.wires.size();
    if (n_public) *n_public = B.n_public;
    return ZK_OK;
}

}  // extern "C"

namespace zkmi {
// uniform-enough field elements for prover randomness / toxic waste (upstream: fr.SetRandom's rejection sampling over crypto/rand)
static int random_frs(HFr* out, int n, bool nonzero) {
    FILE* f = fopen("/dev/urandom", "rb");
    if (!f) return set_err(ZK_ERR_ARG, "no randomness source");
    for (int i = 0; i < n; i++) {
        uint64_t t[4];
        do {
            if (fread(t, 1, 32, f) != 32) { fclose(f); return set_err(ZK_ERR_ARG, "no randomness source"); }
            t[3] &= 0x3fffffffffffffffULL;
        } while (HFr::geq_mod(t) || (nonzero && !(t[0] | t[1] | t[2] | t[3])));  // rejection: r > 2^253, at most ~1.3 draws on average
        out[i] = HFr{{t[0], t[1], t[2], t[3]}}.to_mont();
    }
    fclose(f);
    return ZK_OK;
}
static void hex_of(const uint8_t* b, size_t n, char* o) {
    static const char* dg = "0123456789abcdef";
    for (size_t i = 0; i < n; i++) { o[2 * i] = dg[b[i] >> 4]; o[2 * i + 1] = dg[b[i] & 15]; }
}

// ------------------------------------------------------------------------------------------------ Groth16: what stays resident between calls
// The intended exports (r1cs.go:74-266) take ONE text per call: the RawR1CS JSON carries the circuit (gates, public inputs: 0.25 GB at 2^20 constraints) AND
// this proof's witness values (a hex string, 64 MB) -- plus, for ProveWithPK, the key text (0.37 GB).  Read naively that is 0.86 s of host and key work in
// front of a 10 ms prover (round 4: profiles/rnd5_a_groth16_export_baseline.json).  The same remedy as for PLONK, adapted to the one-text payload:
//   RawCircuit  a RawR1CS text MINUS its values string: identified by the content keys of the bytes before and after that string.  A later call is the
//               same circuit iff the values string sits at the same offset with the same length and both outside parts hash the same -- a string of hex
//               digits (the decoder on the device rejects anything else) cannot change what the tokenizer makes of the rest.  Kept: the R1CS in HBM (CSR),
//               the witness -> wire order, the operand wires of every product variable, buffers for the values text, the wire vector and a, b, c.
//               The wire vector of a proof is then assembled on the device: decode the values (wire.hip), gather (public first), one product per mul term.
//   G16Key      a key text by its content key -> the decoded resident key.  No window tables on a key's first proof (0.15 s at 2^20 against the 5 ms per proof
//               they save); the second proof with it builds them.
// both least-recently-used within ZKMI_TABLE_CAP_GB together with the PLONK entries above.
__global__ void k_raw_wires(const Fr* __restrict__ vals, const uint32_t* __restrict__ order, size_t n_values, Fr one, Fr* __restrict__ w) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_values) return;
    w[i] = i == 0 ? one : vals[order[i - 1]];
}
__global__ void k_raw_products(const uint32_t* __restrict__ pa, const uint32_t* __restrict__ pb, size_t n_products, size_t base, Fr* __restrict__ w) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_products) return;
    w[base + j] = w[pa[j]] * w[pb[j]];  // operands are witness wires (below `base`), never products: no ordering among the lanes
}

struct RawCircuit {
    ContentKey head, tail;
    size_t head_len = 0, tail_len = 0, span_len = 0;  // the text = head | values string (span_len characters) | tail
    int entry = 0;
    size_t n_values = 0, n_wires = 0, n_public = 0, n_constraints = 0, n_products = 0;
    std::vector<uint32_t> public_order;  // witness (0-based) behind public wire 1 + k
    std::unique_ptr<RawR1CSBuilt> host;  // until the circuit is uploaded
    std::mutex work;                     // one proof at a time per circuit
    uint64_t r1cs = 0;
    uint32_t *d_order = nullptr, *d_pa = nullptr, *d_pb = nullptr;
    char* d_text = nullptr;  // 8 bytes past a 16-byte boundary: the felts start aligned (wire.hip)
    void *d_vals = nullptr, *d_w = nullptr, *d_abc = nullptr;
    size_t dev_bytes() const { return n_values * (4 + 64 + 32) + n_products * 8 + n_wires * 32 + n_constraints * (96 + 3 * 4 + 200); }
    // what the accounting reads (any thread, under g_cache_mu) while device_buffers() (under `work` only) swaps the host form for the device form: two flags,
    // the device's set BEFORE the host's is cleared, instead of the unique_ptr and the handle themselves
    std::atomic<bool> has_host{true}, has_dev{false};
    size_t bytes() const { return public_order.size() * 4 + (has_host.load() ? dev_bytes() / 2 : 0) + (has_dev.load() ? dev_bytes() : 0); }
    ~RawCircuit() {
        for (void* q : {(void*)d_order, (void*)d_pa, (void*)d_pb, (void*)d_text, d_vals, d_w, d_abc})
            if (q) (void)hipFree(q);
        if (r1cs) (void)zk_bn254_r1cs_free(r1cs);
    }
    int device_buffers() {  // under `work`
        if (r1cs) return ZK_OK;
        ZK_TRY(ensure_init());
        if (!host) return set_err(ZK_ERR_ARG, "resident circuit lost its host form");
        uint64_t h = 0;
        Phase ph;
        ZK_TRY(r1cs_of_built(*host, &h));
        ph.lap("export.circuit_r1cs_load");
        auto up = [](uint32_t** d, const RawVec<uint32_t>& v) -> int {
            ZK_HIP(hipMalloc((void**)d, (v.size() ? v.size() : 1) * 4));
            if (!v.empty()) ZK_TRY(h2d_big(*d, v.data(), v.size() * 4, nullptr));
            return ZK_OK;
        };
        int rc = up(&d_order, host->order);
        if (rc == ZK_OK) rc = up(&d_pa, host->prod_a);
        if (rc == ZK_OK) rc = up(&d_pb, host->prod_b);
        if (rc == ZK_OK && hipStreamSynchronize(nullptr) != hipSuccess) rc = set_err(ZK_ERR_HIP, "circuit upload failed");
        ph.lap("export.circuit_order_uploads");
        auto dm = [](void** d, size_t bytes) -> int { ZK_HIP(hipMalloc(d, bytes ? bytes : 16)); return ZK_OK; };
        if (rc == ZK_OK) rc = dm((void**)&d_text, span_len + 32);
        if (rc == ZK_OK) rc = dm(&d_vals, n_values * 32);
        if (rc == ZK_OK) rc = dm(&d_w, n_wires * 32);
        if (rc == ZK_OK) rc = dm(&d_abc, 3 * n_constraints * 32);
        ph.lap("export.circuit_scratch_allocs");
        if (rc != ZK_OK) {  // nothing half-made stays behind: a later call starts over
            (void)zk_bn254_r1cs_free(h);
            for (void* q : {(void*)d_order, (void*)d_pa, (void*)d_pb, (void*)d_text, d_vals, d_w, d_abc})
                if (q) (void)hipFree(q);
            d_order = d_pa = d_pb = nullptr;
            d_text = nullptr;
            d_vals = d_w = d_abc = nullptr;
            return rc;
        }
        r1cs = h;
        has_dev.store(true);
        has_host.store(false);
        // the host form (0.3 GB at 2^20 constraints) is given back to the system on a thread of its own: 25 ms this call need not wait for
        std::thread([form = host.release()] { delete form; }).detach();
        return ZK_OK;
    }
};
struct G16Key {
    ContentKey pk;
    uint64_t handle = 0;
    size_t bytes = 0, text_len = 0;  // text_len: characters of the key text this entry was read from
    int in_use = 0;
    unsigned proofs = 0;
};
// (never destroyed: at process exit the R1CS registry of r1cs.hip and the HIP runtime may be gone before a static destructor here would free into them)
static std::list<std::shared_ptr<RawCircuit>>& g_raw = *new std::list<std::shared_ptr<RawCircuit>>();  // most recently used first; under g_cache_mu
static std::list<G16Key>& g_g16_keys = *new std::list<G16Key>();
static constexpr unsigned kG16TablesAtProof = 2;

static size_t g16_cache_bytes_locked() {
    size_t t = 0;
    for (auto& k : g_g16_keys) t += k.bytes;
    for (auto& c : g_raw) t += c->bytes();
    return t;
}
static size_t plonk_cache_bytes_locked() {
    size_t t = 0;
    for (auto& k : g_keys) t += k.bytes;
    for (auto& l : g_lowered) t += l->bytes();
    return t;
}
// room for `extra` more bytes among the Groth16 entries (idle ones, least recently used first).  PLONK's entries are trimmed by their own rule, but their bytes
// count here as these count there: the two caches together stay within the one documented bound (ZKMI_TABLE_CAP_GB)
static void g16_trim_locked(size_t extra, std::vector<uint64_t>* dead_keys) {
    const size_t cap = cache_cap_bytes();
    extra += plonk_cache_bytes_locked();
    for (auto it = g_g16_keys.end(); it != g_g16_keys.begin() && (g16_cache_bytes_locked() + extra > cap || g_g16_keys.size() >= 8);) {
        --it;
        if (it->in_use) continue;
        dead_keys->push_back(it->handle);
        it = g_g16_keys.erase(it);
    }
    while (!g_raw.empty() && (g16_cache_bytes_locked() + extra > cap || g_raw.size() >= 8)) {
        if (g_raw.back().use_count() > 1) break;
        g_raw.pop_back();
    }
}
static void g16_free_keys(const std::vector<uint64_t>& hs) {
    for (uint64_t h : hs) (void)zk_bn254_groth16_pk_free(h);
}

// the circuit behind a RawR1CS text: from the cache (*values_at = where this text keeps its values string), or read now
// `pending` (optional): with exactly one resident circuit that fits this text structurally, return it AT ONCE and let the content keys be compared on other
// threads -- *pending then says whether it really is this text's circuit; the caller may meanwhile do anything that only touches the candidate's scratch buffers
// (upload, decode and assemble the values: 1 of the 3 ms the keys take at 2^20 constraints) and must look at *pending before it uses the circuit itself.
static int raw_circuit_get(const char* raw, size_t len, std::shared_ptr<RawCircuit>* out, size_t* values_at, std::future<bool>* pending = nullptr) {
    Phase ph;
    // candidates: the resident circuits whose span fits this text -- quotes at both ends, the same count header
    std::vector<std::shared_ptr<RawCircuit>> cand;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (auto& c : g_raw)
            if (c->entry == current_entry() && c->head_len + c->span_len + c->tail_len == len && c->head_len >= 1 && raw[c->head_len - 1] == '"' &&
                raw[c->head_len + c->span_len] == '"')
                cand.push_back(c);
    }
    if (pending && cand.size() == 1) {
        std::shared_ptr<RawCircuit> c = cand[0];
        size_t n = 0;
        if (count_from_hex(raw + c->head_len, c->span_len, &n) && n == c->n_values) {
            *pending = std::async(std::launch::async, [raw, c] {
                const double t0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
                std::future<ContentKey> tail = std::async(std::launch::async, [raw, c] { return content_key(raw + c->head_len + c->span_len, c->tail_len); });
                const ContentKey head = content_key(raw, c->head_len);
                const bool same = tail.get() == c->tail && head == c->head;
                prof_host("export.raw_content_key", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0);
                return same;
            });
            {
                std::lock_guard<std::mutex> lk(g_cache_mu);
                for (auto it = g_raw.begin(); it != g_raw.end(); ++it)
                    if (it->get() == c.get()) { g_raw.splice(g_raw.begin(), g_raw, it); break; }
            }
            *out = c;
            *values_at = c->head_len;
            return ZK_OK;
        }
    }
    for (auto& c : cand) {
        size_t n = 0;
        if (!count_from_hex(raw + c->head_len, c->span_len, &n) || n != c->n_values) continue;
        std::future<ContentKey> tail = std::async(std::launch::async, [&] { return content_key(raw + c->head_len + c->span_len, c->tail_len); });
        const ContentKey head = content_key(raw, c->head_len);
        if (!(tail.get() == c->tail) || !(head == c->head)) continue;
        ph.lap("export.raw_content_key");
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (auto it = g_raw.begin(); it != g_raw.end(); ++it)
            if (it->get() == c.get()) { g_raw.splice(g_raw.begin(), g_raw, it); break; }
        *out = c;
        *values_at = c->head_len;
        return ZK_OK;
    }
    auto C = std::make_shared<RawCircuit>();
    C->host.reset(new RawR1CSBuilt());
    {
        std::string err;
        const int rc = raw_r1cs_build(raw, len, C->host.get(), &err, false);
        if (rc != ZK_OK) return set_err(rc, "%s", err.c_str());
    }
    ph.lap("export.raw_parse_lower");
    const RawR1CSBuilt& B = *C->host;
    C->entry = current_entry();
    C->n_values = B.n_values;
    C->n_products = B.prod_a.size();
    C->n_wires = 1 + B.n_values + B.prod_a.size();
    C->n_public = B.n_public;
    C->n_constraints = B.ptr[0].size() - 1;
    C->public_order.assign(B.order.begin(), B.order.begin() + (B.n_public - 1));
    *out = C;
    *values_at = B.values_at;
    if (!B.values_at || !cache_cap_bytes()) { C->span_len = B.values_len; return ZK_OK; }  // (a values string with escapes has no span in the text: never cached)
    C->head_len = B.values_at;
    C->span_len = B.values_len;
    C->tail_len = len - B.values_at - B.values_len;
    {
        std::future<ContentKey> tail = std::async(std::launch::async, [&] { return content_key(raw + C->head_len + C->span_len, C->tail_len); });
        C->head = content_key(raw, C->head_len);
        C->tail = tail.get();
    }
    ph.lap("export.raw_content_key");
    std::vector<uint64_t> dead;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        g16_trim_locked(C->dev_bytes(), &dead);
        g_raw.push_front(C);
    }
    g16_free_keys(dead);
    return ZK_OK;
}

// the wire vector of this proof in C->d_w (under C->work): values text -> device, decoded, gathered public-first, one product per mul term
static int raw_wires_on_device(RawCircuit& C, const char* values, size_t values_len) {
    Phase ph;
    ZK_TRY(C.device_buffers());
    ph.lap("export.circuit_to_device");
    if (values_len != C.span_len) return set_err(ZK_ERR_LEN, "felt vector: %zu characters, the circuit was read with %zu", values_len, C.span_len);
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = g.s->stream;
    char* d_text = C.d_text + 8;
    ZK_HIP(hipMemcpyAsync(d_text, values, values_len, hipMemcpyHostToDevice, st));
    ph.lap("export.values_upload");
    // on the slot this call holds (no second one); the status word: the 4 bytes in front of the text's padding.  Synchronises; ZK_ERR_ARG on a non-hex or non-canonical felt
    ZK_TRY(felts_decode_hex_on_slot(g.s, st, d_text, values_len, C.d_vals, C.n_values, C.n_values, 1, (int*)C.d_text));
    ph.lap("export.values_decode");
    Fr one;
    {
        const HFr h1 = HFr::one();
        memcpy(&one, &h1, 32);
    }
    ZK_LAUNCH(g.s, st, "raw_wires", k_raw_wires, dim3((unsigned)((C.n_values + 256) / 256)), dim3(256), 0, (const Fr*)C.d_vals, (const uint32_t*)C.d_order, C.n_values, one,
              (Fr*)C.d_w);
    if (C.n_products)
        ZK_LAUNCH(g.s, st, "raw_products", k_raw_products, dim3((unsigned)((C.n_products + 255) / 256)), dim3(256), 0, (const uint32_t*)C.d_pa, (const uint32_t*)C.d_pb,
                  C.n_products, 1 + C.n_values, (Fr*)C.d_w);
    ZK_TRY(slot_sync(g.s, st));
    ph.lap("export.witness_assemble");
    return ZK_OK;
}

// the resident key behind a key text (or decoded now, without window tables); *cached: it is pinned in the cache until g16_key_release
static int g16_key_get(const char* pk_hex, size_t pk_len, const ContentKey* known, uint64_t* handle, bool* cached) {
    Phase ph;
    const ContentKey key = known ? *known : content_key(pk_hex, pk_len);
    ph.lap("export.pk_content_key");
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (auto it = g_g16_keys.begin(); it != g_g16_keys.end(); ++it)
            if (it->pk == key && hentry(it->handle) == current_entry()) {
                it->in_use++;
                *handle = it->handle;
                *cached = true;
                g_g16_keys.splice(g_g16_keys.begin(), g_g16_keys, it);
                return ZK_OK;
            }
    }
    *cached = false;
    uint64_t h = 0;
    int rc = zk_bn254_groth16_pk_read(pk_hex, pk_len, 1, 1, 0, &h);
    if (rc == ZK_ERR_HIP) {  // out of HBM with idle keys resident: let them go and try once more
        std::vector<uint64_t> dead;
        {
            std::lock_guard<std::mutex> lk(g_cache_mu);
            for (auto it = g_g16_keys.begin(); it != g_g16_keys.end();)
                if (!it->in_use) { dead.push_back(it->handle); it = g_g16_keys.erase(it); } else ++it;
        }
        if (!dead.empty()) {
            g16_free_keys(dead);
            rc = zk_bn254_groth16_pk_read(pk_hex, pk_len, 1, 1, 0, &h);
        }
    }
    ZK_TRY(rc);
    ph.lap("export.pk_read");
    *handle = h;
    size_t bytes = 0;
    (void)zk_bn254_groth16_pk_bytes(h, &bytes);
    if (!cache_cap_bytes() || bytes > cache_cap_bytes()) return ZK_OK;  // not kept: the caller frees it
    std::vector<uint64_t> dead;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        g16_trim_locked(bytes, &dead);
        G16Key e;
        e.pk = key; e.handle = h; e.bytes = bytes; e.in_use = 1; e.text_len = pk_len;
        g_g16_keys.push_front(e);
        *cached = true;
    }
    g16_free_keys(dead);
    return ZK_OK;
}
static void g16_key_release(uint64_t handle);
// the window tables (and the session's high-priority streams) of a resident key, built off the calling thread; see zk_groth16_prove_with_pk
static void g16_tables_in_background(uint64_t h) {
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        bool found = false;
        for (auto& k : g_g16_keys)
            if (k.handle == h) { k.in_use++; found = true; break; }
        if (!found) return;
    }
    static const bool sync_build = ZK_EXP("ZKMI_EXPORT_TABLES_SYNC", 0) != 0;  // experiment: the round-5 behaviour (build inside the second call)
    auto job = [h] {
        Phase ph;
        CtxScope sc(hentry(h));
        if (sc.rc == ZK_OK && !bg_cancelled()) {
            int built = 0;
            if (zk_bn254_groth16_pk_build_tables(h, 0, &built) == ZK_OK) {
                size_t bytes = 0;
                (void)zk_bn254_groth16_pk_bytes(h, &bytes);
                std::lock_guard<std::mutex> lk(g_cache_mu);
                for (auto& k : g_g16_keys)
                    if (k.handle == h) k.bytes = bytes;
            }
            ph.lap("export.pk_window_tables");
        }
        g16_key_release(h);
    };
    if (sync_build) {
        Phase ph;
        (void)zk_warm_session_streams();
        ph.lap("export.session_streams");
        job();
        return;
    }
    (void)zk_warm_session_streams_background();
    bg_submit(job);
}
static void g16_key_release(uint64_t handle) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    for (auto& k : g_g16_keys)
        if (k.handle == handle) { if (k.in_use > 0) k.in_use--; return; }
}
// a key Setup has just produced enters the cache under the content key of the text it was written as
static bool g16_key_adopt(const char* pk_hex, size_t pk_len, uint64_t h) {
    size_t bytes = 0;
    if (zk_bn254_groth16_pk_bytes(h, &bytes) != ZK_OK || !cache_cap_bytes() || bytes > cache_cap_bytes()) return false;
    const ContentKey key = content_key(pk_hex, pk_len);
    std::vector<uint64_t> dead;
    bool adopted = false;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        bool have = false;
        for (auto& k : g_g16_keys) have = have || (k.pk == key && hentry(k.handle) == current_entry());
        if (!have) {
            g16_trim_locked(bytes, &dead);
            G16Key e;
            e.pk = key; e.handle = h; e.bytes = bytes; e.in_use = 0; e.text_len = pk_len;
            g_g16_keys.push_front(e);
            adopted = true;
        }
    }
    g16_free_keys(dead);
    return adopted;
}

// One key waits here between the two calls of a Preprocess -- the size query and the call that writes it -- so that Setup runs once (and, with drawn toxic
// waste, so that the sizes reported ARE those of the key that gets written).
struct G16Stash {
    std::shared_ptr<RawCircuit> circuit;
    bool drawn = false;
    HFr toxic[5];
    uint64_t handle = 0;
    std::vector<zk_g1_affine> vk_g1;
    zk_g2_affine vk_g2[3];
};
static G16Stash& g_g16_stash = *new G16Stash();

// prove on a resident circuit with a resident key (under C.work; C.d_w holds this proof's wires)
static int g16_prove_resident(RawCircuit& C, uint64_t pk, const zk_fr* rs, uint8_t proof[128]) {
    Phase ph;
    {   // the key must be the key of THIS circuit's shape
        size_t kw = 0, kp = 0;
        uint32_t lg = 0;
        ZK_TRY(zk_bn254_groth16_pk_info(pk, &kw, &kp, &lg, nullptr));
        if (kw != C.n_wires || kp != C.n_public || C.n_constraints > ((size_t)1 << lg))
            return set_err(ZK_ERR_ARG, "proving key is for %zu wires / %zu public / 2^%u constraints, the circuit has %zu / %zu / %zu", kw, kp, lg, C.n_wires, C.n_public, C.n_constraints);
    }
    HFr r2[2];
    if (rs) memcpy(r2, rs, sizeof r2);
    else ZK_TRY(random_frs(r2, 2, false));
    Fr* abc = (Fr*)C.d_abc;
    const size_t nc = C.n_constraints;
    ZK_TRY(zk_bn254_r1cs_eval_abc_dev(C.r1cs, C.d_w, C.n_wires, abc, abc + nc, abc + 2 * nc, nullptr));
    ph.lap("export.r1cs_solve_abc");
    ZK_TRY(zk_bn254_groth16_prove(pk, abc, abc + nc, abc + 2 * nc, nc, C.d_w, C.n_wires, (const zk_fr*)&r2[0], (const zk_fr*)&r2[1], 1, proof));
    ph.lap("export.groth16_prove");
    return ZK_OK;
}

}  // namespace zkmi

extern "C" {

// Reads a RawR1CS text into the resident cache; to_device = 0 touches no device (the export shim runs it beside the HIP runtime's start), 1 also uploads the
// circuit (the shim's second step, beside the key's decoding).  The zk_groth16_* call that follows finds the circuit by content.
// The values span of a RawR1CS text is identified by its position and its count header when the circuit is found resident, and raw_r1cs_build(want_wires = false)
// does not decode it either: the entry points that never hand the span to the device decoder (Preprocess, the lowering alone) check here that every character is
// one hex.DecodeString accepts -- as the reference's DeserializeFelts does on every call -- so that a span with a '"' and further JSON members in it cannot
// pass for the resident circuit.  Eight threads over 64 MB at 2^20 constraints: a few milliseconds.
static int raw_span_is_hex(const char* span, size_t n) {
    const unsigned T = n >= ((size_t)1 << 20) ? 8 : 1;
    std::vector<std::future<bool>> part;
    for (unsigned t = 1; t < T; t++) part.push_back(std::async(std::launch::async, [=] { return all_hex(span + n * t / T, n * (t + 1) / T - n * t / T); }));
    bool ok = all_hex(span, n / T);
    for (auto& f : part) ok = f.get() && ok;
    return ok ? ZK_OK : set_err(ZK_ERR_ARG, "felt vector: invalid hex character");
}
int zk_groth16_lower_resident(const char* raw_json, size_t raw_len, int to_device) {
    if (!raw_json) return set_err(ZK_ERR_ARG, "null pointer");
    std::shared_ptr<RawCircuit> C;
    size_t at = 0;
    ZK_TRY(raw_circuit_get(raw_json, raw_len, &C, &at));
    if (!to_device) return at ? raw_span_is_hex(raw_json + at, C->span_len) : ZK_OK;
    ZK_TRY(ensure_init());
    std::lock_guard<std::mutex> work(C->work);
    Phase ph;
    ZK_TRY(C->device_buffers());
    ph.lap("export.circuit_to_device");
    return ZK_OK;
}
// Decodes a key text into the resident cache (no window tables: see zk_groth16_prove_with_pk); the ProveWithPK that follows finds it by content key.
int zk_groth16_key_resident(const char* pk_hex, size_t pk_len) {
    if (!pk_hex) return set_err(ZK_ERR_ARG, "null pointer");
    ZK_TRY(ensure_init());
    uint64_t h = 0;
    bool cached = false;
    ZK_TRY(g16_key_get(pk_hex, pk_len, nullptr, &h, &cached));
    if (cached) g16_key_release(h);
    else (void)zk_bn254_groth16_pk_free(h);
    return ZK_OK;
}
// The public witness of a RawR1CS payload as the verifier needs it (buildWitnesses' public part, r1cs.go:176-212): the values of the public wires after ONE,
// in wire order.  Host only (a process that only verifies never starts the HIP runtime); uses the resident circuit of the text when there is one.
// *n_public comes back even when cap is too small (ZK_ERR_ARG then).
int zk_groth16_public_inputs(const char* raw_json, size_t raw_len, zk_fr* out, size_t cap, size_t* n_public) {
    if (!raw_json || !n_public) return set_err(ZK_ERR_ARG, "null pointer");
    std::shared_ptr<RawCircuit> C;
    size_t at = 0;
    ZK_TRY(raw_circuit_get(raw_json, raw_len, &C, &at));
    const size_t np = C->public_order.size();
    *n_public = np;
    if (np > cap || (np && !out)) return set_err(ZK_ERR_ARG, "%zu public inputs, the output holds %zu", np, cap);
    if (!at) {  // a values string with escapes: the general reader
        RawR1CSBuilt B;
        std::string err;
        const int rc = raw_r1cs_build(raw_json, raw_len, &B, &err);
        if (rc != ZK_OK) return set_err(rc, "%s", err.c_str());
        for (size_t k = 0; k < np; k++) memcpy(&out[k], &B.wires[1 + k], 32);
        return ZK_OK;
    }
    // DeserializeFelts sees the whole vector: one bad character anywhere fails the call (hex.DecodeString), and so does a non-canonical felt
    const char* v = raw_json + at;
    if (!all_hex(v, C->span_len)) return set_err(ZK_ERR_ARG, "felt vector: invalid hex character");
    for (size_t k = 0; k < np; k++) {
        uint64_t t[4] = {0, 0, 0, 0};
        const char* s = v + 8 + 64 * (size_t)C->public_order[k];
        for (int i = 0; i < 64; i++) t[i >> 4] |= (uint64_t)(hex_nibble((unsigned char)s[63 - i]) & 15) << (4 * (i & 15));
        if (HFr::geq_mod(t)) return set_err(ZK_ERR_ARG, "felt vector: invalid fr.Element encoding (value >= r)");
        HFr m = HFr{{t[0], t[1], t[2], t[3]}}.to_mont();
        memcpy(&out[k], &m, 32);
    }
    return ZK_OK;
}

// Preprocess of the reference's intended Groth16 FFI (backend/groth16/r1cs.go:214-266): RawR1CS JSON -> groth16.Setup -> hex(ProvingKey.WriteTo),
// hex(VerifyingKey.WriteTo).  toxic: tau, alpha, beta, gamma, delta (Montgomery, non-zero) or NULL (/dev/urandom, as upstream draws them).
// pk_hex_out == NULL: only the sizes (the key is built to learn NbInfinityA / NbInfinityB -- and waits for the call that writes it).  pk_handle (optional)
// keeps the key resident for the caller; without it the key enters the export cache under the text it was written as.
int zk_groth16_preprocess(const char* raw_json, size_t raw_len, const zk_fr* toxic, char* pk_hex_out, size_t pk_cap, size_t* pk_len, char* vk_hex_out, size_t vk_cap,
                          size_t* vk_len, uint64_t* pk_handle) {
    InFlight _in_flight;
    if (!raw_json || !pk_len || !vk_len) return set_err(ZK_ERR_ARG, "null pointer");
    std::shared_ptr<RawCircuit> C;
    size_t at = 0;
    ZK_TRY(raw_circuit_get(raw_json, raw_len, &C, &at));
    if (at) ZK_TRY(raw_span_is_hex(raw_json + at, C->span_len));  // Setup does not read the values, the reference's DeserializeFelts still refuses a bad string
    ZK_TRY(ensure_init());
    std::lock_guard<std::mutex> work(C->work);
    Phase ph;
    ZK_TRY(C->device_buffers());
    ph.lap("export.circuit_to_device");
    HFr tx[5];
    if (toxic) memcpy(tx, toxic, sizeof tx);
    uint64_t h = 0;
    std::vector<zk_g1_affine> vk_g1;
    zk_g2_affine vk_g2[3];
    {   // the key the size query built, if this is the call that follows it (same circuit, same -- or equally drawn -- toxic waste)
        std::lock_guard<std::mutex> lk(g_cache_mu);
        G16Stash& S = g_g16_stash;
        if (S.handle && S.circuit.get() == C.get() && (toxic ? (!S.drawn && !memcmp(S.toxic, tx, sizeof tx)) : S.drawn)) {
            h = S.handle;
            vk_g1.swap(S.vk_g1);
            memcpy(vk_g2, S.vk_g2, sizeof vk_g2);
            S.handle = 0;
            S.circuit.reset();
        }
    }
    if (!h) {
        if (!toxic) ZK_TRY(random_frs(tx, 5, true));
        vk_g1.resize(1 + C->n_public);
        ZK_TRY(zk_bn254_groth16_setup(C->r1cs, (const zk_fr*)tx, 1, &h, vk_g1.data(), vk_g2));  // no window tables: the prover that keeps the key builds them
        ph.lap("export.groth16_setup");
    }
    int rc = zk_bn254_groth16_pk_write(h, 1, pk_hex_out, pk_cap, pk_len);
    if (rc == ZK_OK) rc = zk_bn254_groth16_vk_write(h, vk_g1.data(), C->n_public, vk_g2, 1, pk_hex_out ? vk_hex_out : nullptr, vk_cap, vk_len);
    if (pk_hex_out) ph.lap("export.pk_write_hex");
    if (rc == ZK_OK && !pk_hex_out) {  // the size query: the key waits for the writing call
        uint64_t old = 0;
        {
            std::lock_guard<std::mutex> lk(g_cache_mu);
            G16Stash& S = g_g16_stash;
            old = S.handle;
            S.handle = h;
            S.circuit = C;
            S.drawn = toxic == nullptr;
            memcpy(S.toxic, tx, sizeof tx);
            S.vk_g1.swap(vk_g1);
            memcpy(S.vk_g2, vk_g2, sizeof vk_g2);
        }
        if (old) (void)zk_bn254_groth16_pk_free(old);
        return ZK_OK;
    }
    if (rc == ZK_OK && pk_handle) { *pk_handle = h; return ZK_OK; }
    if (rc != ZK_OK || !g16_key_adopt(pk_hex_out, *pk_len, h)) (void)zk_bn254_groth16_pk_free(h);
    ph.lap("export.pk_adopt");
    return rc;
}

// ProveWithPK (r1cs.go:107-143): RawR1CS JSON + hex(ProvingKey.WriteTo) -> hex(Proof.WriteTo) (256 characters, no terminator).  pk_hex may be NULL when
// pk_handle names a resident key (the reference deserialises the key on every call).  rs: the prover's (r, s) or NULL (/dev/urandom).
// Both texts are found resident by content when they were seen before; a key read from its text gets its window tables when its SECOND proof is asked for.
int zk_groth16_prove_with_pk(const char* raw_json, size_t raw_len, const char* pk_hex, size_t pk_len, uint64_t pk_handle, const zk_fr* rs, char proof_hex_out[256]) {
    InFlight _in_flight;
    if (!raw_json || !proof_hex_out || (!pk_hex && !pk_handle)) return set_err(ZK_ERR_ARG, "null pointer");
    CtxScope _scope(pk_handle ? hentry(pk_handle) : current_entry());
    if (_scope.rc != ZK_OK) return _scope.rc;
    // the key text's content key beside the circuit's (0.37 GB at 2^20 constraints: 2 ms of sixteen threads); a future of std::async joins in its destructor,
    // so every early return below leaves the caller's text alone
    std::future<ContentKey> pk_key;
    if (pk_hex) pk_key = std::async(std::launch::async, [pk_hex, pk_len] { return content_key(pk_hex, pk_len); });
    std::shared_ptr<RawCircuit> C;
    size_t at = 0;
    std::future<bool> same_circuit;  // (joins in its destructor: an early return leaves the caller's text alone)
    ZK_TRY(raw_circuit_get(raw_json, raw_len, &C, &at, &same_circuit));
    ZK_TRY(ensure_init());
    std::unique_lock<std::mutex> work(C->work);
    ContentKey pk_ck;
    bool have_pk_ck = false;
    if (same_circuit.valid()) {
        // a resident circuit fits this text and its content keys are being compared: this proof's values go to the device meanwhile (they only touch scratch buffers)
        const int rw = raw_wires_on_device(*C, raw_json + at, C->span_len);
        // ... and so does the PROOF, when a resident key that already has its window tables was read from a text of this length: 3.5 ms of hashing (0.63 GB of
        // text at 2^20 constraints) run beside the 10 ms prover instead of in front of it.  The proof leaves this call only if BOTH comparisons say "same".
        uint64_t spec = 0;
        ContentKey spec_ck;
        static const bool speculate = ZK_EXP("ZKMI_EXPORT_SPECULATE", 1) != 0;
        if (speculate && pk_hex && rw == ZK_OK) {
            std::lock_guard<std::mutex> lk(g_cache_mu);
            for (auto& k : g_g16_keys)
                if (k.text_len == pk_len && k.proofs >= kG16TablesAtProof && hentry(k.handle) == current_entry()) {
                    k.in_use++;
                    spec = k.handle;
                    spec_ck = k.pk;
                    break;
                }
        }
        uint8_t proof[128];
        const int rp = spec ? g16_prove_resident(*C, spec, rs, proof) : ZK_ERR_ARG;
        const bool same = same_circuit.get();
        if (spec) {
            pk_ck = pk_key.get();
            have_pk_ck = true;
            const bool ours = same && rp == ZK_OK && pk_ck == spec_ck;
            {
                std::lock_guard<std::mutex> lk(g_cache_mu);
                for (auto it = g_g16_keys.begin(); it != g_g16_keys.end(); ++it)
                    if (it->handle == spec) {
                        it->in_use--;
                        if (ours) { it->proofs++; g_g16_keys.splice(g_g16_keys.begin(), g_g16_keys, it); }
                        break;
                    }
            }
            if (ours) {
                hex_of(proof, 128, proof_hex_out);
                return ZK_OK;
            }
        }
        if (!same) {  // another circuit after all: read the text
            work.unlock();
            C.reset();
            ZK_TRY(raw_circuit_get(raw_json, raw_len, &C, &at));
            work = std::unique_lock<std::mutex>(C->work);
            if (at) ZK_TRY(raw_wires_on_device(*C, raw_json + at, C->span_len));
        } else {
            ZK_TRY(rw);
        }
    } else if (at) {
        ZK_TRY(raw_wires_on_device(*C, raw_json + at, C->span_len));
    }
    if (!at) {  // a values string written with escapes has no view in the text: the general reader decodes it and runs the solver's step on the host
        RawR1CSBuilt B;
        std::string err;
        const int rb = raw_r1cs_build(raw_json, raw_len, &B, &err);
        if (rb != ZK_OK) return set_err(rb, "%s", err.c_str());
        ZK_TRY(C->device_buffers());
        ZK_HIP(hipMemcpy(C->d_w, B.wires.data(), B.wires.size() * 32, hipMemcpyHostToDevice));
    }
    uint64_t h = pk_handle;
    bool cached = false;
    if (pk_hex) {
        if (!have_pk_ck) pk_ck = pk_key.get();
        ZK_TRY(g16_key_get(pk_hex, pk_len, &pk_ck, &h, &cached));
    }
    struct Done {
        uint64_t h; bool from_text, cached;
        ~Done() { if (from_text) { if (cached) g16_key_release(h); else (void)zk_bn254_groth16_pk_free(h); } }
    } done{h, pk_hex != nullptr, cached};
    if (cached) {
        bool build = false;
        {
            std::lock_guard<std::mutex> lk(g_cache_mu);
            for (auto& k : g_g16_keys)
                if (k.handle == h) { build = ++k.proofs == kG16TablesAtProof; break; }
        }
        // A key that proves again is worth its window tables (113 ms to build at 2^20 constraints, 5 ms saved per proof) and the schedule's high-priority streams
        // (39 ms; a lean process has none until now) -- but not on THIS call's critical path: both go to the library's background thread (ctx.hip) and this proof
        // runs without them, as the key's first one did.  The tables are published under the key registry's mutex when they are complete; proofs that start
        // after that find them.  The cache entry stays pinned (in_use) while the build reads the key's base arrays.
        if (build) g16_tables_in_background(h);
    }
    uint8_t proof[128];
    ZK_TRY(g16_prove_resident(*C, h, rs, proof));
    hex_of(proof, 128, proof_hex_out);
    return ZK_OK;
}

// ProveWithMeta (r1cs.go:74-105): Setup and Prove in one call (the proving key never leaves HBM).
int zk_groth16_prove_with_meta(const char* raw_json, size_t raw_len, const zk_fr* toxic, const zk_fr* rs, char proof_hex_out[256]) {
    InFlight _in_flight;
    if (!raw_json || !proof_hex_out) return set_err(ZK_ERR_ARG, "null pointer");
    std::shared_ptr<RawCircuit> C;
    size_t at = 0;
    ZK_TRY(raw_circuit_get(raw_json, raw_len, &C, &at));
    ZK_TRY(ensure_init());
    std::lock_guard<std::mutex> work(C->work);
    if (!at) return set_err(ZK_ERR_ARG, "RawR1CS JSON: the values string must be plain hex (no escapes)");
    ZK_TRY(raw_wires_on_device(*C, raw_json + at, C->span_len));
    HFr tx[5];
    if (toxic) memcpy(tx, toxic, sizeof tx);
    else ZK_TRY(random_frs(tx, 5, true));
    uint64_t h = 0;
    ZK_TRY(zk_bn254_groth16_setup(C->r1cs, (const zk_fr*)tx, 1, &h, nullptr, nullptr));  // one proof: window tables would cost more than they save
    uint8_t proof[128];
    const int rc = g16_prove_resident(*C, h, rs, proof);
    (void)zk_bn254_groth16_pk_free(h);
    if (rc == ZK_OK) hex_of(proof, 128, proof_hex_out);
    return rc;
}

// Releases everything the export path keeps resident between calls (lowered circuits, decoded proving keys; PLONK and Groth16).  Keys in use by a running proof stay.
int zk_export_cache_clear(void) {
    (void)zk_background_wait(-1);  // a background table build pins its key: let it finish so that the key can go
    std::vector<uint64_t> dead, dead16;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (auto it = g_keys.begin(); it != g_keys.end();)
            if (!it->in_use) { dead.push_back(it->handle); it = g_keys.erase(it); } else ++it;
        g_lowered.clear();
        for (auto it = g_g16_keys.begin(); it != g_g16_keys.end();)
            if (!it->in_use) { dead16.push_back(it->handle); it = g_g16_keys.erase(it); } else ++it;
        g_raw.clear();
        if (g_g16_stash.handle) dead16.push_back(g_g16_stash.handle);
        g_g16_stash.handle = 0;
        g_g16_stash.circuit.reset();
    }
    free_handles(dead);
    g16_free_keys(dead16);
    return ZK_OK;
}
// resident entries and their HBM + host bytes (tests, bench.py)
int zk_export_cache_info(size_t* n_circuits, size_t* n_keys, size_t* bytes) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    size_t t = g16_cache_bytes_locked();
    for (auto& k : g_keys) t += k.bytes;
    for (auto& l : g_lowered) t += l->bytes();
    if (n_circuits) *n_circuits = g_lowered.size() + g_raw.size();
    if (n_keys) *n_keys = g_keys.size() + g_g16_keys.size();
    if (bytes) *bytes = t;
    return ZK_OK;
}

// The number of G1 points of an SRS the export shim CREATES when <config dir>/noir-lang/srs.hex is missing: the reference's 1,000,000 (backend/common.go:137)
// unless a test asked for fewer (they take seconds to generate and minutes to read back in the Python oracle).  4 .. 2^28; anything else is ZK_ERR_ARG.
static size_t g_new_srs_size = 1000000;
int zk_export_set_new_srs_size(size_t n) {
    if (n < 4 || n > ((size_t)1 << 28)) return set_err(ZK_ERR_ARG, "SRS size %zu outside [4, 2^28]", n);
    g_new_srs_size = n;
    return ZK_OK;
}
size_t zk_export_new_srs_size(void) { return g_new_srs_size; }

// The lowering alone, for inspection / tests: gates of an ACIR circuit as the reference's BuildSparseR1CS emits them.  Any out pointer may be
// NULL; arrays need *n_constraints (first call with NULL arrays to size them) entries; coefficients come back as Montgomery fr.Elements.
int zk_acir_to_sparse_r1cs(const char* acir_json, size_t acir_len, size_t n_values, int layout, size_t* n_public, size_t* n_vars, size_t* n_constraints, zk_fr* ql,
                           zk_fr* qr, zk_fr* qo, zk_fr* qm, zk_fr* qk, uint32_t* xa, uint32_t* xb, uint32_t* xc, uint32_t* order) {
    if (!acir_json) return set_err(ZK_ERR_ARG, "null pointer");
    Gates G;
    ZK_TRY(lower(acir_json, acir_len, n_values, layout, true, &G));
    if (n_public) *n_public = G.n_public;
    if (n_vars) *n_vars = G.n_vars;
    if (n_constraints) *n_constraints = G.xa.size();
    const size_t nc = G.xa.size();
    if (ql) memcpy(ql, G.ql.data(), nc * 32);
    if (qr) memcpy(qr, G.qr.data(), nc * 32);
    if (qo) memcpy(qo, G.qo.data(), nc * 32);
    if (qm) memcpy(qm, G.qm.data(), nc * 32);
    if (qk) memcpy(qk, G.qk.data(), nc * 32);
    if (xa) memcpy(xa, G.xa.data(), nc * 4);
    if (xb) memcpy(xb, G.xb.data(), nc * 4);
    if (xc) memcpy(xc, G.xc.data(), nc * 4);
    if (order) memcpy(order, G.order.data(), G.n_vars * 4);
    return ZK_OK;
}

}  // extern "C"
