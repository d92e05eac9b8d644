// Host-only readers of the untrusted byte / hex images that cross the reference's FFI (no HIP in this file: tests/cpp/parser_fuzz.cpp compiles it
// with plain g++ under -fsanitize=address,undefined).  Everything here decides from the caller's bytes ALONE whether an image is well formed and how
// large its sections are -- before a single byte goes to the device -- so that a hostile length field can neither wrap an offset nor size an
// allocation (the reference log.Fatal()s on any of these: gnark_backend_ffi/main.go:26-30,46-50,61-72; internal/backend/helpers.go:24-33,49-94).
//   felt vectors          hex( u32 BE count | count x 32 B BE )                         src/gnark_backend_wrapper/serialize.rs:33-47
//   kzg.SRS.WriteTo       G2[0] | G2[1] (64 B compressed each) | u32 BE count | count x 32 B compressed G1       backend/common.go:86-125
//   plonk.ProvingKey      VerifyingKey 368 B | 2 x Domain 168 B | 9 x (u32 BE n | n x 32 B) | 3n x int64 BE      internal/backend/helpers.go:49-60,82-87
//   groth16.ProvingKey    Domain 168 B | 3 x G1 | A, B, Z, K slices | 2 x G2 | G2.B slice | nbWires, NbInfinityA, NbInfinityB u64 | 2 x nbWires bools
//                                                                                                                 backend/groth16/r1cs.go:118-128,214-266
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/zkmi.h"
#include "host_ff.hpp"

namespace zkmi {

static inline int hexv(int c) { return (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : (c >= 'A' && c <= 'F') ? c - 'A' + 10 : -1; }

struct Blob {  // the caller's buffer: bytes, or their hex text
    const uint8_t* p;
    size_t len;
    bool hex;
    size_t nbytes() const { return hex ? len / 2 : len; }
    bool get(size_t off, size_t n, uint8_t* dst) const {
        if (off > nbytes() || n > nbytes() - off) return false;
        if (!hex) { memcpy(dst, p + off, n); return true; }
        for (size_t i = 0; i < n; i++) {
            const int h = hexv(p[2 * (off + i)]), l = hexv(p[2 * (off + i) + 1]);
            if (h < 0 || l < 0) return false;
            dst[i] = (uint8_t)((h << 4) | l);
        }
        return true;
    }
    bool u32(size_t off, size_t* v) const {
        uint8_t b[4];
        if (!get(off, 4, b)) return false;
        *v = ((size_t)b[0] << 24) | ((size_t)b[1] << 16) | ((size_t)b[2] << 8) | b[3];
        return true;
    }
    bool u64(size_t off, uint64_t* v) const {
        uint8_t b[8];
        if (!get(off, 8, b)) return false;
        *v = 0;
        for (int i = 0; i < 8; i++) *v = (*v << 8) | b[i];
        return true;
    }
};

static inline std::string fmt_err(const char* fmt, unsigned long long a = 0, unsigned long long b = 0, unsigned long long c = 0, unsigned long long d = 0) {
    char m[320];
    snprintf(m, sizeof m, fmt, a, b, c, d);
    return m;
}

// ---- the shim's text helpers (goffi.cpp)
// encodedValues reach PlonkPreprocess as a JSON string (main.go:66-72: "TODO: Fix this in the Rust backend side") and the other exports bare: a view
// without the quotes, no copy (the witness vector of a 2^19-gate circuit is 33 MB of text)
static inline void unquote(const char* p, size_t n, const char** out, size_t* out_n) {
    if (n >= 2 && p[0] == '"' && p[n - 1] == '"') { p++; n -= 2; }
    *out = p;
    *out_n = n;
}
// is every character one hex.DecodeString accepts?  (the witness vector of a 2^19-gate circuit is 33 MB of text: a table and no early exit per character)
static inline bool all_hex(const char* p, size_t n) {
    static const struct Tab {
        uint8_t bad[256];
        Tab() { for (int c = 0; c < 256; c++) bad[c] = hexv(c) < 0; }
    } T;
    unsigned acc = 0;
    size_t i = 0;
    for (; i + 8 <= n; i += 8)
        acc |= T.bad[(uint8_t)p[i]] | T.bad[(uint8_t)p[i + 1]] | T.bad[(uint8_t)p[i + 2]] | T.bad[(uint8_t)p[i + 3]] | T.bad[(uint8_t)p[i + 4]] | T.bad[(uint8_t)p[i + 5]] |
               T.bad[(uint8_t)p[i + 6]] | T.bad[(uint8_t)p[i + 7]];
    for (; i < n; i++) acc |= T.bad[(uint8_t)p[i]];
    return acc == 0;
}
static inline bool hex_to_bytes(const char* h, size_t n, std::vector<uint8_t>* out) {
    if (n & 1) return false;
    out->resize(n / 2);
    for (size_t i = 0; i < out->size(); i++) {
        const int a = hexv((unsigned char)h[2 * i]), b = hexv((unsigned char)h[2 * i + 1]);
        if (a < 0 || b < 0) return false;
        (*out)[i] = (uint8_t)((a << 4) | b);
    }
    return true;
}
// DeserializeFelts on the host for the handful of public inputs a verifier needs: u32 BE count | count x 32 B BE -> canonical big-endian elements
static inline bool felts_from_hex(const char* h, size_t n, std::vector<std::vector<uint8_t>>* out) {
    std::vector<uint8_t> b;
    if (!hex_to_bytes(h, n, &b) || b.size() < 4) return false;
    const size_t cnt = ((size_t)b[0] << 24) | ((size_t)b[1] << 16) | ((size_t)b[2] << 8) | b[3];
    if ((b.size() - 4) / 32 != cnt || (b.size() - 4) % 32) return false;
    out->clear();
    for (size_t i = 0; i < cnt; i++) out->emplace_back(b.begin() + 4 + 32 * i, b.begin() + 36 + 32 * i);
    return true;
}
// canonical big-endian 32-byte elements -> Montgomery images (value < r required, like fr.Vector.UnmarshalBinary)
static inline bool be_to_mont(const std::vector<std::vector<uint8_t>>& be, std::vector<zk_fr>* out) {
    out->resize(be.size());
    for (size_t k = 0; k < be.size(); k++) {
        if (be[k].size() != 32) return false;
        uint64_t t[4];
        for (int i = 0; i < 4; i++) {
            uint64_t v = 0;
            for (int b = 0; b < 8; b++) v = (v << 8) | be[k][8 * (3 - i) + b];
            t[i] = v;
        }
        if (HFr::geq_mod(t)) return false;
        const HFr m = HFr{{t[0], t[1], t[2], t[3]}}.to_mont();
        memcpy(&(*out)[k], &m, 32);
    }
    return true;
}

// ---- plonk.ProvingKey.ReadFrom: the header and every length prefix
static const size_t PLONK_PK_HEAD = 368 + 2 * 168;
struct PlonkKeyHeader {
    size_t n = 0, n_public = 0, nbytes = 0;
    unsigned logn = 0, logN4 = 0;
    uint8_t head[PLONK_PK_HEAD];
};
// Key headers come from untrusted bytes (PlonkProveWithPK hands over whatever the caller sends): NbPublicVariables must fit the variables the
// circuit has -- the prover reads that many elements of the solution -- and neither count may be large enough to wrap a sum.
static inline int plonk_check_counts(uint64_t n_public, size_t n_vars, size_t n_constraints, std::string* err) {
    if (n_public > n_vars) { *err = fmt_err("proving key: %llu public inputs but %llu variables", n_public, n_vars); return ZK_ERR_ARG; }
    if (n_public >= ((uint64_t)1 << 28) || n_constraints >= ((size_t)1 << 28)) {
        *err = fmt_err("proving key: %llu public inputs + %llu constraints exceed the Fr two-adicity 2^28", n_public, n_constraints);
        return ZK_ERR_ARG;
    }
    return ZK_OK;
}
static inline int plonk_domains_for(size_t size_system, unsigned* logn, unsigned* logN4, std::string* err) {
    if (size_system < 2) { *err = "PLONK needs at least 2 rows (constraints + public inputs)"; return ZK_ERR_ARG; }
    unsigned ln = 0;
    while (((size_t)1 << ln) < size_system) ln++;
    const size_t big = (size_system < 6 ? 8 : 4) * size_system;
    unsigned lb = 0;
    while (((size_t)1 << lb) < big) lb++;
    if (lb > 28) { *err = fmt_err("PLONK big domain 2^%llu exceeds the Fr two-adicity 2^28", lb); return ZK_ERR_ARG; }
    *logn = ln;
    *logN4 = lb;
    return ZK_OK;
}
static inline int plonk_pk_header(const void* data, size_t len, int is_hex, size_t n_vars, size_t n_constraints, PlonkKeyHeader* H, std::string* err) {
    if (is_hex && (len & 1)) { *err = "proving key: odd number of hex characters"; return ZK_ERR_LEN; }
    const Blob B{(const uint8_t*)data, len, is_hex != 0};
    const size_t nbytes = B.nbytes();
    if (nbytes < PLONK_PK_HEAD) { *err = fmt_err("proving key: truncated (%llu bytes, %llu needed)", nbytes, PLONK_PK_HEAD); return ZK_ERR_LEN; }
    if (!B.get(0, PLONK_PK_HEAD, H->head)) { *err = "proving key: invalid hex character"; return ZK_ERR_ARG; }
    auto be64 = [](const uint8_t* p) { uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | p[i]; return v; };
    const uint64_t size = be64(H->head), npub = be64(H->head + 72), card0 = be64(H->head + 368), card1 = be64(H->head + 368 + 168);
    int rc = plonk_check_counts(npub, n_vars, n_constraints, err);
    if (rc != ZK_OK) return rc;
    rc = plonk_domains_for(n_constraints + (size_t)npub, &H->logn, &H->logN4, err);
    if (rc != ZK_OK) return rc;
    if (size != ((uint64_t)1 << H->logn) || card0 != size || card1 != ((uint64_t)1 << H->logN4)) {
        char m[320];
        snprintf(m, sizeof m, "proving key: domain sizes %llu / %llu / %llu do not match %zu constraints + %llu public inputs", (unsigned long long)size,
                 (unsigned long long)card0, (unsigned long long)card1, n_constraints, (unsigned long long)npub);
        *err = m;
        return ZK_ERR_ARG;
    }
    const size_t n = (size_t)size;  // <= 2^28: none of the products below can wrap
    const size_t want = PLONK_PK_HEAD + 9 * (4 + 32 * n) + 24 * n;
    if (nbytes != want) { *err = fmt_err("proving key: %llu bytes, %llu expected for a domain of %llu", nbytes, want, n); return ZK_ERR_LEN; }
    for (int k = 0; k < 9; k++) {
        size_t pre = 0;
        if (!B.u32(PLONK_PK_HEAD + (size_t)k * (4 + 32 * n), &pre)) { *err = "proving key: invalid hex character"; return ZK_ERR_ARG; }
        if (pre != n) { *err = fmt_err("proving key: vector %llu does not hold %llu elements", (unsigned long long)k, n); return ZK_ERR_LEN; }
    }
    if (is_hex && (nbytes & 3)) { *err = "proving key: length is not a multiple of 4 bytes"; return ZK_ERR_LEN; }
    H->n = n;
    H->n_public = (size_t)npub;
    H->nbytes = nbytes;
    return ZK_OK;
}

// ---- kzg.SRS.ReadFrom: the two G2 points and the count
struct SrsHeader {
    size_t n_g1 = 0, nbytes = 0;
    uint8_t g2[128];
};
static inline int kzg_srs_header(const void* data, size_t len, int is_hex, SrsHeader* H, std::string* err) {
    const Blob B{(const uint8_t*)data, len, is_hex != 0};
    const size_t nbytes = B.nbytes();
    if ((is_hex && (len & 1)) || nbytes < 132) { *err = fmt_err("SRS: %llu bytes cannot hold the two G2 points and the count", nbytes); return ZK_ERR_LEN; }
    if (nbytes & 3) { *err = fmt_err("SRS: %llu bytes, the count cannot match", nbytes); return ZK_ERR_LEN; }
    size_t n = 0;
    if (!B.get(0, 128, H->g2) || !B.u32(128, &n)) { *err = "SRS: invalid hex character"; return ZK_ERR_ARG; }
    if ((nbytes - 132) / 32 != n || (nbytes - 132) % 32) { *err = fmt_err("SRS: %llu bytes, the count says %llu G1 points (%llu bytes)", nbytes, n, 132 + 32 * (unsigned long long)n); return ZK_ERR_LEN; }
    H->n_g1 = n;
    H->nbytes = nbytes;
    return ZK_OK;
}

// ---- groth16.ProvingKey.ReadFrom: the section table
struct Groth16KeyHeader {
    uint64_t card = 0, n_wires = 0, nb_inf_a = 0, nb_inf_b = 0;
    unsigned logN = 0;
    size_t cnt[5] = {0, 0, 0, 0, 0}, at[5] = {0, 0, 0, 0, 0};  // A, B, Z, K, G2.B: point counts and byte offsets of the first point
    size_t at_g2 = 0, at_bitmaps = 0, total_bytes = 0;
    uint8_t domain[168];
    std::vector<uint8_t> inf_a, inf_b;
};
static inline int groth16_pk_header(const void* data, size_t len, int is_hex, Groth16KeyHeader* H, std::string* err) {
    if (is_hex && (len & 1)) { *err = "proving key: odd number of hex characters"; return ZK_ERR_LEN; }
    const Blob B{(const uint8_t*)data, len, is_hex != 0};
    const size_t nbytes = B.nbytes();
    const char* trunc = "proving key: truncated or invalid hex in the header fields";
    if (!B.u64(0, &H->card)) { *err = trunc; return ZK_ERR_LEN; }
    unsigned logN = 0;
    while (logN < 28 && ((uint64_t)1 << logN) < H->card) logN++;
    if (H->card == 0 || ((uint64_t)1 << logN) != H->card) { *err = fmt_err("proving key: domain cardinality %llu is not a power of two <= 2^28", H->card); return ZK_ERR_ARG; }
    H->logN = logN;
    size_t off = 168 + 96;
    for (int k = 0; k < 4; k++) {
        if (!B.u32(off, &H->cnt[k])) { *err = trunc; return ZK_ERR_LEN; }
        H->at[k] = off + 4;
        if (H->cnt[k] > (nbytes - H->at[k]) / 32) { *err = fmt_err("proving key: %llu bytes cannot hold a slice of %llu G1 points", nbytes, H->cnt[k]); return ZK_ERR_LEN; }
        off = H->at[k] + 32 * H->cnt[k];
    }
    H->at_g2 = off;
    if (nbytes - off < 128 + 4) { *err = trunc; return ZK_ERR_LEN; }
    off += 128;
    if (!B.u32(off, &H->cnt[4])) { *err = trunc; return ZK_ERR_LEN; }
    H->at[4] = off + 4;
    if (H->cnt[4] > (nbytes - H->at[4]) / 64) { *err = fmt_err("proving key: %llu bytes cannot hold a slice of %llu G2 points", nbytes, H->cnt[4]); return ZK_ERR_LEN; }
    off = H->at[4] + 64 * H->cnt[4];
    if (nbytes - off < 24) { *err = trunc; return ZK_ERR_LEN; }
    if (!B.u64(off, &H->n_wires) || !B.u64(off + 8, &H->nb_inf_a) || !B.u64(off + 16, &H->nb_inf_b)) { *err = trunc; return ZK_ERR_LEN; }
    H->at_bitmaps = off + 24;
    const uint64_t nw = H->n_wires, nia = H->nb_inf_a, nib = H->nb_inf_b;
    if (nw >= ((uint64_t)1 << 31) || nbytes != H->at_bitmaps + 2 * (size_t)nw) {
        *err = fmt_err("proving key: %llu bytes, the fields say %llu wires (%llu bytes)", nbytes, nw, H->at_bitmaps + 2 * (unsigned long long)nw);
        return ZK_ERR_LEN;
    }
    if (nia > nw || nib > nw || H->cnt[0] != nw - nia || H->cnt[1] != nw - nib || H->cnt[4] != nw - nib) {
        *err = "proving key: the point counts do not match NbInfinityA / NbInfinityB";
        return ZK_ERR_ARG;
    }
    if (H->cnt[2] != H->card || H->cnt[3] > nw) {
        *err = fmt_err("proving key: Z holds %llu points for a domain of %llu, K %llu for %llu wires", H->cnt[2], H->card, H->cnt[3], nw);
        return ZK_ERR_ARG;
    }
    H->inf_a.assign(nw ? (size_t)nw : 1, 0);
    H->inf_b.assign(nw ? (size_t)nw : 1, 0);
    if (!B.get(H->at_bitmaps, (size_t)nw, H->inf_a.data()) || !B.get(H->at_bitmaps + (size_t)nw, (size_t)nw, H->inf_b.data())) { *err = "proving key: invalid hex character"; return ZK_ERR_ARG; }
    for (size_t i = 0; i < nw; i++)
        if (H->inf_a[i] > 1 || H->inf_b[i] > 1) { *err = "proving key: a bool that is neither 0 nor 1"; return ZK_ERR_ARG; }
    if (!B.get(0, 168, H->domain)) { *err = trunc; return ZK_ERR_LEN; }
    H->total_bytes = nbytes;
    return ZK_OK;
}

}  // namespace zkmi
