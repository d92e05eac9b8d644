// Key / SRS wire formats on the device -- SURVEY §8 row f1: what the reference moves between Rust and Go as hex strings and keeps in
// srs.hex  [REF gnark_backend_ffi/internal/backend/helpers.go:49-94 (Serialize/DeserializeProvingKey, VerifyingKey, Proof: hex of gnark's
// WriteTo bytes), backend/common.go:86-125 (LoadSRS / SaveSRS: hex(kzg.SRS.WriteTo) re-read from disk on EVERY prove / verify call,
// backend/plonk/plonk.go:16,34,58)].  Encodings are gnark-crypto v0.9.1's ecc/bn254/marshal.go  [UPSTREAM-RECALL]: integers big-endian,
// fr / fp elements 32 B big-endian canonical, points COMPRESSED (G1: X with two flag bits 0b10 / 0b11 = smallest / largest Y, 0b01 =
// infinity), slices prefixed by a u32 big-endian length.
// The point of doing it here: a compressed point costs a square root (one 254-bit exponentiation: y = (x^3 + 3)^((q+1)/4)), 10^6 of
// them per SRS load on the CPU path; on the device the decoded points land directly in the resident-bases layout (and its window tables)
// and the SRS is read ONCE.  Byte / integer work next to ~380 field products per point.
#include <string.h>

#include <vector>

#include "ctx.hpp"
#include "ff.hpp"
#include "keyio.hpp"
#include "msm.hpp"

namespace zkmi {

// ---- hex
__device__ __forceinline__ uint32_t hexdig4(uint32_t w, uint32_t* bad) {  // 4 characters -> 2 bytes (text order, low byte first); see wire.hip hex4
    uint32_t nib = (w & 0x0f0f0f0fu) + ((w >> 6) & 0x01010101u) * 9u;
    uint32_t gt9 = ((nib + 0x06060606u) >> 4) & 0x01010101u;
    uint32_t enc = nib + 0x30303030u + gt9 * 0x27u;
    *bad |= (enc ^ (w | ((w >> 1) & 0x20202020u))) | (nib & 0xf0f0f0f0u);
    uint32_t b = ((nib << 4) | (nib >> 8)) & 0x00ff00ffu;
    return (b & 0xffu) | ((b >> 8) & 0xff00u);
}
// one lane: 8 characters -> 4 bytes
__global__ void k_hex_decode(const uint32_t* __restrict__ text, size_t n_words, uint32_t* __restrict__ out, int* __restrict__ status) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_words) return;
    uint32_t bad = 0;
    uint32_t lo = hexdig4(text[2 * i], &bad), hi = hexdig4(text[2 * i + 1], &bad);
    out[i] = lo | (hi << 16);
    if (bad) atomicOr(status, 1);
}
__device__ __forceinline__ uint32_t hexenc2w(uint32_t b16) {
    uint32_t nib = ((b16 >> 4) & 0x0fu) | ((b16 & 0x0fu) << 8) | (((b16 >> 12) & 0x0fu) << 16) | (((b16 >> 8) & 0x0fu) << 24);
    uint32_t gt9 = ((nib + 0x06060606u) >> 4) & 0x01010101u;
    return nib + 0x30303030u + gt9 * 0x27u;
}
__global__ void k_hex_encode(const uint32_t* __restrict__ bytes, size_t n_words, uint32_t* __restrict__ text) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_words) return;
    uint32_t w = bytes[i];
    text[2 * i] = hexenc2w(w & 0xffffu);
    text[2 * i + 1] = hexenc2w(w >> 16);
}

// ---- 32-byte big-endian integers <-> 8 little-endian limbs, through 4-byte loads (the vectors inside a key sit at any 4-byte offset)
template <class F>
__device__ __forceinline__ F load_be32(const uint32_t* p) {
    F x;
#pragma unroll
    for (int k = 0; k < 8; k++) x.l[7 - k] = __builtin_bswap32(p[k]);
    return x;
}
template <class F>
__device__ __forceinline__ void store_be32(uint32_t* p, const F& x) {
#pragma unroll
    for (int k = 0; k < 8; k++) p[k] = __builtin_bswap32(x.l[7 - k]);
}
template <class P>
__device__ __forceinline__ bool geq_mod(const uint32_t x[8]) {
    for (int i = 7; i >= 0; i--)
        if (x[i] != P::MOD[i]) return x[i] > P::MOD[i];
    return true;
}
__global__ void k_fr_from_be(const uint32_t* __restrict__ raw, size_t n, Fr* __restrict__ out, int* __restrict__ status) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr x = load_be32<Fr>(raw + 8 * i);
    if (geq_mod<FrParams>(x.l)) { atomicOr(status, 2); return; }  // gnark-crypto: "invalid fr.Element encoding"
    out[i] = x.to_mont();
}
__global__ void k_fr_to_be(const Fr* __restrict__ in, size_t n, uint32_t* __restrict__ raw) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    store_be32(raw + 8 * i, in[i].from_mont());
}

// ---- G1 points
__device__ __forceinline__ bool fp_lex_largest_dev(const Fp& canonical) {  // value > (q - 1) / 2
    for (int i = 7; i >= 0; i--) {
        uint32_t h = (FpParams::MOD[i] >> 1) | (i < 7 ? FpParams::MOD[i + 1] << 31 : 0);
        if (canonical.l[i] != h) return canonical.l[i] > h;
    }
    return false;
}
// G1Affine.SetBytes on a compressed encoding: y = sqrt(x^3 + 3) = (x^3 + 3)^((q + 1) / 4)  (q = 3 mod 4), sign by the flag
__global__ __launch_bounds__(256) void k_g1_decompress(const uint32_t* __restrict__ raw, size_t n, Affine<Fp>* __restrict__ out, int* __restrict__ status) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp x = load_be32<Fp>(raw + 8 * i);
    const uint32_t flag = x.l[7] >> 30;
    x.l[7] &= 0x3fffffffu;
    Affine<Fp> p = Affine<Fp>::inf();
    if (flag == 1) {  // infinity: the rest must be zero
        if (!x.is_zero()) atomicOr(status, 4);
        out[i] = p;
        return;
    }
    if (flag == 0 || geq_mod<FpParams>(x.l)) {  // an uncompressed encoding inside a compressed slice / x >= q
        atomicOr(status, 4);
        out[i] = p;
        return;
    }
    Fp xm = x.to_mont();
    Fp three = Fp::one() + Fp::one() + Fp::one();
    Fp rhs = xm.sqr() * xm + three;
    // (q + 1) / 4
    const uint32_t e[8] = {0xb61f3f52u, 0x4f082305u, 0x5a1c72a3u, 0x65e05aa4u, 0xa0605617u, 0x6e14116du, 0xb84c680au, 0x0c19139cu};
    Fp y = rhs.pow(e);
    if (y.sqr() != rhs) {  // not on the curve
        atomicOr(status, 4);
        out[i] = p;
        return;
    }
    if (fp_lex_largest_dev(y.from_mont()) != (flag == 3)) y = Fp::zero() - y;
    p.x = xm;
    p.y = y;
    out[i] = p;
}
__global__ void k_g1_compress(const Affine<Fp>* __restrict__ pts, size_t n, uint32_t* __restrict__ raw) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<Fp> p = pts[i];
    Fp x = Fp::zero();
    uint32_t flag = 1;
    if (!p.is_inf()) {
        x = p.x.from_mont();
        flag = fp_lex_largest_dev(p.y.from_mont()) ? 3 : 2;
    }
    x.l[7] |= flag << 30;
    store_be32(raw + 8 * i, x);
}

static unsigned grid1(size_t n) { return (unsigned)((n + 255) / 256); }

int hex_decode_dev(Slot* s, hipStream_t st, const void* d_text, size_t n_bytes, void* d_out, int* d_status) {
    if (n_bytes & 3) return set_err(ZK_ERR_ARG, "hex payload of %zu bytes is not a multiple of 4", n_bytes);
    if (n_bytes) ZK_LAUNCH(s, st, "hex_decode", k_hex_decode, dim3(grid1(n_bytes / 4)), dim3(256), 0, (const uint32_t*)d_text, n_bytes / 4, (uint32_t*)d_out, d_status);
    return ZK_OK;
}
int hex_encode_dev(Slot* s, hipStream_t st, const void* d_bytes, size_t n_bytes, void* d_text) {
    if (n_bytes & 3) return set_err(ZK_ERR_ARG, "payload of %zu bytes is not a multiple of 4", n_bytes);
    if (n_bytes) ZK_LAUNCH(s, st, "hex_encode", k_hex_encode, dim3(grid1(n_bytes / 4)), dim3(256), 0, (const uint32_t*)d_bytes, n_bytes / 4, (uint32_t*)d_text);
    return ZK_OK;
}
int fr_from_be_dev(Slot* s, hipStream_t st, const void* d_raw, size_t n, void* d_out, int* d_status) {
    if (n) ZK_LAUNCH(s, st, "fr_from_be", k_fr_from_be, dim3(grid1(n)), dim3(256), 0, (const uint32_t*)d_raw, n, (Fr*)d_out, d_status);
    return ZK_OK;
}
int fr_to_be_dev(Slot* s, hipStream_t st, const void* d_in, size_t n, void* d_raw) {
    if (n) ZK_LAUNCH(s, st, "fr_to_be", k_fr_to_be, dim3(grid1(n)), dim3(256), 0, (const Fr*)d_in, n, (uint32_t*)d_raw);
    return ZK_OK;
}
int g1_decompress_dev(Slot* s, hipStream_t st, const void* d_raw, size_t n, void* d_out, int* d_status) {
    if (n) ZK_LAUNCH(s, st, "g1_decompress", k_g1_decompress, dim3(grid1(n)), dim3(256), 0, (const uint32_t*)d_raw, n, (Affine<Fp>*)d_out, d_status);
    return ZK_OK;
}
int g1_compress_dev(Slot* s, hipStream_t st, const void* d_pts, size_t n, void* d_raw) {
    if (n) ZK_LAUNCH(s, st, "g1_compress", k_g1_compress, dim3(grid1(n)), dim3(256), 0, (const Affine<Fp>*)d_pts, n, (uint32_t*)d_raw);
    return ZK_OK;
}

// ---- G2 on the host (two points per SRS)
static HFp2 f2_pow(HFp2 a, const uint64_t e[4]) {
    HFp2 r = HFp2::one();
    for (int i = 0; i < 256; i++) {
        if ((e[i >> 6] >> (i & 63)) & 1) r = r * a;
        a = a.sqr();
    }
    return r;
}
// square root in Fp2 = Fp[u]/(u^2 + 1), q = 3 mod 4 (Adj & Rodriguez-Henriquez, Alg. 9)
static bool f2_sqrt(const HFp2& a, HFp2* out) {
    if (a.is_zero()) { *out = a; return true; }
    // (q - 3) / 4 and (q - 1) / 2
    static const uint64_t E1[4] = {0x4f082305b61f3f51ULL, 0x65e05aa45a1c72a3ULL, 0x6e14116da0605617ULL, 0x0c19139cb84c680aULL};
    static const uint64_t E2[4] = {0x9e10460b6c3e7ea3ULL, 0xcbc0b548b438e546ULL, 0xdc2822db40c0ac2eULL, 0x183227397098d014ULL};
    const HFp2 minus_one = HFp2{HFp::zero() - HFp::one(), HFp::zero()};
    HFp2 a1 = f2_pow(a, E1);
    HFp2 alpha = a1 * (a1 * a);
    HFp2 a0 = HFp2{alpha.a0, alpha.a1.neg()} * alpha;
    if (a0 == minus_one) return false;
    HFp2 x0 = a1 * a;
    if (alpha == minus_one) {
        *out = HFp2{HFp::zero(), HFp::one()} * x0;
    } else {
        HFp2 b = f2_pow(HFp2::one() + alpha, E2);
        *out = b * x0;
    }
    return out->sqr() == a;
}
static bool fp_lex_largest_host(const HFp& mont) {
    HFp c = mont.from_mont();
    uint64_t h[4];
    for (int i = 0; i < 4; i++) h[i] = (HFpParams::MOD[i] >> 1) | (i < 3 ? HFpParams::MOD[i + 1] << 63 : 0);
    for (int i = 3; i >= 0; i--)
        if (c.l[i] != h[i]) return c.l[i] > h[i];
    return false;
}
static bool fp_from_be(const uint8_t in[32], HFp* out) {
    uint64_t t[4];
    for (int i = 0; i < 4; i++) {
        uint64_t v = 0;
        for (int b = 0; b < 8; b++) v = (v << 8) | in[8 * (3 - i) + b];
        t[i] = v;
    }
    if (HFp::geq_mod(t)) return false;
    *out = HFp{{t[0], t[1], t[2], t[3]}}.to_mont();
    return true;
}
bool g2_decompress_host(const uint8_t in[64], Affine<HFp2>* out) {
    const unsigned flag = in[0] >> 6;
    *out = Affine<HFp2>::inf();
    if (flag == 1) return true;
    if (flag == 0) return false;
    uint8_t b[64];
    memcpy(b, in, 64);
    b[0] &= 0x3f;
    HFp2 x;
    if (!fp_from_be(b, &x.a1) || !fp_from_be(b + 32, &x.a0)) return false;
    // twist: y^2 = x^3 + 3 / (9 + u)
    HFp nine = HFp::zero(), three = HFp::one() + HFp::one() + HFp::one();
    for (int i = 0; i < 3; i++) nine = nine + three;
    HFp2 bt = HFp2{three, HFp::zero()} * HFp2{nine, HFp::one()}.inv();
    HFp2 rhs = x.sqr() * x + bt, y;
    if (!f2_sqrt(rhs, &y)) return false;
    bool largest = y.a1.is_zero() ? fp_lex_largest_host(y.a0) : fp_lex_largest_host(y.a1);
    if (largest != (flag == 3)) y = y.neg();
    out->x = x;
    out->y = y;
    // subgroup check: r * P == infinity (G2 has a cofactor)
    uint32_t rk[8];
    memcpy(rk, HFrParams::MOD, 32);
    return scalar_mul(*out, rk).is_inf();
}

}  // namespace zkmi

using namespace zkmi;

extern "C" {

// kzg.SRS.ReadFrom on the bytes (or their hex text) of kzg.SRS.WriteTo: G2[0] | G2[1] (64 B compressed each) | u32 BE count | count x 32 B
// compressed G1.  The G1 points are decompressed on the device and registered as resident bases (window tables per table_window_bits as in
// zk_bn254_bases_register_cfg); handle / n_g1 / the two G2 points come back.  This is what replaces LoadSRS (backend/common.go:86-105).
int zk_bn254_kzg_srs_read(const void* data, size_t len, int is_hex, int table_window_bits, uint64_t* handle, size_t* n_g1, zk_g2_affine g2_out[2]) {
    if (!data || !handle) return set_err(ZK_ERR_ARG, "null pointer");
    const size_t nbytes = is_hex ? len / 2 : len;
    if ((is_hex && (len & 1)) || nbytes < 132) return set_err(ZK_ERR_LEN, "SRS: %zu bytes cannot hold the two G2 points and the count", nbytes);
    if (nbytes & 3) return set_err(ZK_ERR_LEN, "SRS: %zu bytes, the count cannot match", nbytes);
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    Slot* s = g.s;
    hipStream_t st = s->stream;
    ZK_TRY(s->reserve(len + nbytes + 2 * nbytes + 65536));
    int* d_status = (int*)s->alloc(64);
    ZK_HIP(hipMemsetAsync(d_status, 0, 4, st));
    uint8_t* d_bytes = (uint8_t*)s->alloc(nbytes + 16);
    if (is_hex) {
        void* d_text = s->alloc(len + 16);
        ZK_HIP(hipMemcpyAsync(d_text, data, len, hipMemcpyHostToDevice, st));
        ZK_TRY(hex_decode_dev(s, st, d_text, nbytes, d_bytes, d_status));
    } else {
        ZK_HIP(hipMemcpyAsync(d_bytes, data, nbytes, hipMemcpyHostToDevice, st));
    }
    uint8_t head[132];
    ZK_HIP(hipMemcpyAsync(head, d_bytes, 132, hipMemcpyDeviceToHost, st));
    ZK_TRY(slot_sync(s, st));
    const size_t n = ((size_t)head[128] << 24) | ((size_t)head[129] << 16) | ((size_t)head[130] << 8) | head[131];
    if (nbytes != 132 + 32 * n) return set_err(ZK_ERR_LEN, "SRS: %zu bytes, the count says %zu G1 points (%zu bytes)", nbytes, n, 132 + 32 * n);
    Affine<HFp2> g2[2];
    for (int k = 0; k < 2; k++)
        if (!g2_decompress_host(head + 64 * k, &g2[k])) return set_err(ZK_ERR_ARG, "SRS: invalid G2 point %d", k);
    void* d_pts = s->alloc(n * 64 + 16);
    ZK_TRY(g1_decompress_dev(s, st, d_bytes + 132, n, d_pts, d_status));
    int h_status = 0;
    ZK_HIP(hipMemcpyAsync(&h_status, d_status, 4, hipMemcpyDeviceToHost, st));
    ZK_TRY(slot_sync(s, st));
    if (h_status & 1) return set_err(ZK_ERR_ARG, "SRS: invalid hex character");
    if (h_status & 4) return set_err(ZK_ERR_ARG, "SRS: invalid compressed G1 point (bad flags, x >= q, or no square root)");
    ZK_TRY(zk_bn254_bases_register_cfg(d_pts, n, 0, 1, table_window_bits, handle));
    if (n_g1) *n_g1 = n;
    if (g2_out) memcpy(g2_out, g2, 256);
    return ZK_OK;
}

// kzg.SRS.WriteTo of a registered G1 base array + the two G2 points: bytes (or hex text) into out; *out_len = bytes written.
int zk_bn254_kzg_srs_write(uint64_t handle, const zk_g2_affine g2[2], int as_hex, void* out, size_t cap, size_t* out_len) {
    if (!g2 || !out || !out_len) return set_err(ZK_ERR_ARG, "null pointer");
    const void* d_pts = nullptr;
    size_t n = 0;
    int is_g2 = 0;
    ZK_TRY(bases_ptr(handle, &d_pts, &n, &is_g2));
    if (is_g2) return set_err(ZK_ERR_ARG, "the SRS handle must be a G1 base array");
    const size_t nbytes = 132 + 32 * n, need = as_hex ? 2 * nbytes : nbytes;
    *out_len = need;
    if (cap < need) return set_err(ZK_ERR_ARG, "output holds %zu bytes, %zu needed", cap, need);
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    Slot* s = g.s;
    hipStream_t st = s->stream;
    ZK_TRY(s->reserve(3 * nbytes + 65536));
    uint8_t* d_bytes = (uint8_t*)s->alloc(nbytes + 16);
    uint8_t head[132];
    Affine<HFp2> gg[2];
    memcpy(gg, g2, 256);
    // G2Affine.Bytes(): X.A1 | X.A0 big-endian, flags on the first byte
    for (int k = 0; k < 2; k++) {
        uint8_t* o = head + 64 * k;
        if (gg[k].is_inf()) { memset(o, 0, 64); o[0] = 0x40; continue; }
        for (int half = 0; half < 2; half++) {
            HFp c = (half ? gg[k].x.a0 : gg[k].x.a1).from_mont();
            for (int i = 0; i < 4; i++)
                for (int b = 0; b < 8; b++) o[32 * half + 31 - (8 * i + b)] = (uint8_t)(c.l[i] >> (8 * b));
        }
        bool largest = gg[k].y.a1.is_zero() ? fp_lex_largest_host(gg[k].y.a0) : fp_lex_largest_host(gg[k].y.a1);
        o[0] |= largest ? 0xC0 : 0x80;
    }
    head[128] = (uint8_t)(n >> 24); head[129] = (uint8_t)(n >> 16); head[130] = (uint8_t)(n >> 8); head[131] = (uint8_t)n;
    ZK_HIP(hipMemcpyAsync(d_bytes, head, 132, hipMemcpyHostToDevice, st));
    ZK_TRY(g1_compress_dev(s, st, d_pts, n, d_bytes + 132));
    if (as_hex) {
        void* d_text = s->alloc(2 * nbytes + 16);
        ZK_TRY(hex_encode_dev(s, st, d_bytes, nbytes, d_text));
        ZK_HIP(hipMemcpyAsync(out, d_text, 2 * nbytes, hipMemcpyDeviceToHost, st));
    } else {
        ZK_HIP(hipMemcpyAsync(out, d_bytes, nbytes, hipMemcpyDeviceToHost, st));
    }
    return slot_sync(s, st);
}

}  // extern "C"
