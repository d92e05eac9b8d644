// Host-side encodings and transcript primitives shared by the provers (groth16.hip, plonk.hip):
//   gnark-crypto v0.9.1 point encodings  G1Affine.Bytes() / RawBytes(), G2Affine.Bytes()      ecc/bn254/marshal.go
//   fr.Element.Marshal() / SetBytes()                                                          ecc/bn254/fr/element.go
//   SHA-256 (FIPS 180-4) -- the hash gnark's PLONK prover hands to fiatshamir.NewTranscript    (plonk prove.go: sha256.New())
// (modules pinned at /root/reference/gnark_backend_ffi/go.mod:5,23; reached through groth16.Prove main.go:131 and
//  plonk.Prove backend/plonk/plonk.go:67).  Layouts are [UPSTREAM-RECALL] -- see DESIGN.md "Oracle and parity".
#pragma once
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "curve.hpp"
#include "host_ff.hpp"

namespace zkmi {

static inline void to_canonical_u32(const HFr& mont, uint32_t out[8]) {
    HFr c = mont.from_mont();
    memcpy(out, c.l, 32);
}

// 32 bytes big-endian canonical
template <class HF>
static inline void field_to_be(const HF& mont, uint8_t out[32]) {
    HF c = mont.from_mont();
    for (int i = 0; i < 4; i++)
        for (int b = 0; b < 8; b++) out[31 - (8 * i + b)] = (uint8_t)(c.l[i] >> (8 * b));
}
static inline void fp_to_be(const HFp& mont, uint8_t out[32]) { field_to_be(mont, out); }
static inline void fr_to_be(const HFr& mont, uint8_t out[32]) { field_to_be(mont, out); }

// fr.Element.SetBytes: big-endian integer, reduced mod r; returns the Montgomery image
static inline HFr fr_from_be_reduce(const uint8_t in[32]) {
    uint64_t t[4];
    for (int i = 0; i < 4; i++) {
        uint64_t v = 0;
        for (int b = 0; b < 8; b++) v = (v << 8) | in[8 * (3 - i) + b];
        t[i] = v;
    }
    while (HFr::geq_mod(t)) HFr::sub_mod(t);  // 2^256 < 6 r
    HFr r{{t[0], t[1], t[2], t[3]}};
    return r.to_mont();
}

static inline bool fp_lex_largest(const HFp& mont) {  // canonical value > (q-1)/2
    HFp c = mont.from_mont();
    uint64_t h[4];
    for (int i = 0; i < 4; i++) h[i] = (HFpParams::MOD[i] >> 1) | (i < 3 ? HFpParams::MOD[i + 1] << 63 : 0);
    for (int i = 3; i >= 0; i--)
        if (c.l[i] != h[i]) return c.l[i] > h[i];
    return false;
}
// gnark-crypto G1Affine.Bytes(): 32 B big-endian X; top bits of byte 0: 10 = y smallest, 11 = y largest, 01 = infinity
static inline void g1_compress(const Affine<HFp>& p, uint8_t out[32]) {
    if (p.is_inf()) {
        memset(out, 0, 32);
        out[0] = 0x40;
        return;
    }
    fp_to_be(p.x, out);
    out[0] |= fp_lex_largest(p.y) ? 0xC0 : 0x80;
}
// G1Affine.RawBytes() / Marshal(): X || Y big-endian, no flag bits for a finite point; infinity = 0x40 then zeros
static inline void g1_raw_bytes(const Affine<HFp>& p, uint8_t out[64]) {
    if (p.is_inf()) {
        memset(out, 0, 64);
        out[0] = 0x40;
        return;
    }
    fp_to_be(p.x, out);
    fp_to_be(p.y, out + 32);
}
static inline void g2_compress(const Affine<HFp2>& p, uint8_t out[64]) {
    if (p.is_inf()) {
        memset(out, 0, 64);
        out[0] = 0x40;
        return;
    }
    fp_to_be(p.x.a1, out);
    fp_to_be(p.x.a0, out + 32);
    bool largest = p.y.a1.is_zero() ? fp_lex_largest(p.y.a0) : fp_lex_largest(p.y.a1);
    out[0] |= largest ? 0xC0 : 0x80;
}

// ---- SHA-256
struct Sha256 {
    uint32_t h[8];
    uint8_t buf[64];
    uint64_t len = 0;
    size_t fill = 0;
    Sha256() { reset(); }
    void reset() {
        static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
        memcpy(h, iv, 32);
        len = 0;
        fill = 0;
    }
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void block(const uint8_t* p) {
        static const uint32_t K[64] = {
            0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
            0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
            0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
            0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
            0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
            0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + K[i] + w[i];
            uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const void* data, size_t n) {
        const uint8_t* p = (const uint8_t*)data;
        len += n;
        while (n) {
            size_t k = 64 - fill < n ? 64 - fill : n;
            memcpy(buf + fill, p, k);
            fill += k; p += k; n -= k;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    void final(uint8_t out[32]) {
        uint64_t bits = len * 8;
        uint8_t pad = 0x80;
        update(&pad, 1);
        uint8_t z = 0;
        while (fill != 56) update(&z, 1);
        uint8_t lb[8];
        for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(lb, 8);
        for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
    }
};

// gnark-crypto fiatshamir.Transcript over SHA-256: challenge_i = H(name_i || challenge_{i-1} || bindings_i...)
struct FsTranscript {
    std::vector<std::string> ids;
    std::vector<std::vector<uint8_t>> bound;   // concatenated bindings per challenge
    std::vector<std::vector<uint8_t>> value;   // 32 bytes once computed
    explicit FsTranscript(std::initializer_list<const char*> names) {
        for (const char* n : names) ids.emplace_back(n);
        bound.resize(ids.size());
        value.resize(ids.size());
    }
    void bind(size_t i, const void* p, size_t n) { bound[i].insert(bound[i].end(), (const uint8_t*)p, (const uint8_t*)p + n); }
    void bind_g1(size_t i, const Affine<HFp>& p) { uint8_t b[64]; g1_raw_bytes(p, b); bind(i, b, 64); }
    void bind_fr(size_t i, const HFr& x) { uint8_t b[32]; fr_to_be(x, b); bind(i, b, 32); }
    // computes challenge i (all previous ones must have been computed, like upstream's errPreviousChallengeNotComputed) -> fr.SetBytes
    HFr challenge(size_t i) {
        Sha256 s;
        s.update(ids[i].data(), ids[i].size());
        if (i) s.update(value[i - 1].data(), value[i - 1].size());
        s.update(bound[i].data(), bound[i].size());
        uint8_t d[32];
        s.final(d);
        value[i].assign(d, d + 32);
        return fr_from_be_reduce(d);
    }
};

}  // namespace zkmi
