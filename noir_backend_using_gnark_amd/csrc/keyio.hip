// Key / SRS wire formats on the device -- SURVEY §8 row f1: what the reference moves between Rust and Go as hex strings and keeps in
// srs.hex  [REF gnark_backend_ffi/internal/backend/helpers.go:49-94 (Serialize/DeserializeProvingKey, VerifyingKey, Proof: hex of gnark's
// WriteTo bytes), backend/common.go:86-125 (LoadSRS / SaveSRS: hex(kzg.SRS.WriteTo) re-read from disk on EVERY prove / verify call,
// backend/plonk/plonk.go:16,34,58)].  Encodings are gnark-crypto v0.9.1's ecc/bn254/marshal.go  [UPSTREAM-RECALL]: integers big-endian,
// fr / fp elements 32 B big-endian canonical, points COMPRESSED (G1: X with two flag bits 0b10 / 0b11 = smallest / largest Y, 0b01 =
// infinity), slices prefixed by a u32 big-endian length.
// The point of doing it here: a compressed point costs a square root (one 254-bit exponentiation: y = (x^3 + 3)^((q+1)/4)), 10^6 of
// them per SRS load on the CPU path; on the device the decoded points land directly in the resident-bases layout (and its window tables)
// and the SRS is read ONCE.  Byte / integer work next to ~380 field products per point.
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <vector>

#include "ctx.hpp"
#include "ff.hpp"
#include "ff29.hpp"
#include "keyio.hpp"
#include "text_host.hpp"
#include "msm.hpp"
#include "ntt.hpp"
#include "proofio.hpp"

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

// ---- square roots: a^((q - 3) / 4) on the 29-bit multiplier (ff29.hpp)
// Both decompressions end in this exponentiation (q = 3 mod 4: sqrt(a) = a^((q+1)/4) = a^((q-3)/4) * a; the Fp2 root takes two).  The saturated
// Field::pow spends 252 squarings + 127 products of ~305 instructions; here the constant exponent is walked in sliding windows of three bits -- 250 squarings
// of ~170 instructions and 55 products of 206 with a, a^3, a^5, a^7 -- 2.1 x fewer instructions.  One byte per window: squarings << 2 | (odd power >> 1),
// most significant window first (the first one only selects the starting power); every value in the chain is a direct product output (< 1.03 p).
__device__ __forceinline__ U29 u29_pow_qm3_4(const U29& a) {
    static const uint8_t W[56] = {9,  29, 12, 16, 23, 23, 22, 9,  20, 17, 21, 8,  42, 17, 19, 30, 24, 26, 14, 14, 14, 33, 38, 13, 30, 9,  13, 22,
                                  15, 38, 14, 18, 12, 26, 14, 31, 27, 22, 8,  21, 8,  23, 4,  20, 24, 21, 34, 14, 14, 4,  31, 9,  23, 15, 18, 16};
    const U29 a2 = u29_sqr(a), a3 = u29_mul(a2, a), a5 = u29_mul(a3, a2), a7 = u29_mul(a5, a2);
    U29 acc = a3;  // W[0] & 3 == 1
#pragma unroll 1
    for (int k = 1; k < 56; k++) {
        const unsigned w = W[k];
#pragma unroll 1
        for (unsigned j = 0; j < (w >> 2); j++) acc = u29_sqr(acc);
        switch (w & 3u) {
            case 0: acc = u29_mul(acc, a); break;
            case 1: acc = u29_mul(acc, a3); break;
            case 2: acc = u29_mul(acc, a5); break;
            default: acc = u29_mul(acc, a7); break;
        }
    }
    return acc;
}
// canonical Montgomery image -> a^((q-3)/4) * a^mul_a as a canonical Montgomery image (mul_a: once more by a, the square-root candidate)
__device__ __forceinline__ Fp fp_pow_qm3_4(const Fp& a, bool times_a) {
    const U29 x = u29_mul(u29_load(a), u29_one());  // contracted: < 1.2 p
    U29 e = u29_pow_qm3_4(x);
    if (times_a) e = u29_mul(e, x);
    return u29_store(e);
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
    Fp y = fp_pow_qm3_4(rhs, true);  // rhs^((q + 1) / 4)
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


// ---- G2 points on the device (a Groth16 proving key holds one per wire)
// Square root in Fp2 = Fp[u]/(u^2 + 1), q = 3 mod 4, by the complex method -- two exponentiations in Fp instead of the two in Fp2 of rounds 2-4 (Adj &
// Rodriguez-Henriquez, Alg. 9: 1,778 base-field products per root; this: ~770).  For a = a0 + a1 u with a1 != 0: the norm n = a0^2 + a1^2 is a square in Fp
// exactly when a is one in Fp2; with s^2 = n and t = (a0 + s) / 2, one exponentiation e = t^((q-3)/4) gives c = e t with c^2 = chi t (chi = +-1 the quadratic
// character of t) AND 1 / c = c e^2 -- no inversion --, and the root is (c, a1 / 2c) if chi = 1, (a1 / 2c, c) otherwise (then c^2 = -t = -(a0 + s) / 2 and
// (a1 / 2c)^2 = (a0 - s) / 2).  Either root will do: the caller picks the sign by the encoding's flag.  `half` = 1 / 2 (Montgomery).
__device__ bool f2_sqrt_dev(const Fp2& a, const Fp& half, Fp2* out) {
    if (a.is_zero()) { *out = a; return true; }
    // every exponentiation below is by (q - 3) / 4: fp_pow_qm3_4
    if (a.a1.is_zero()) {  // a in Fp: sqrt(a0) or u sqrt(-a0)
        const Fp c = fp_pow_qm3_4(a.a0, true);
        if (c.sqr() == a.a0) *out = Fp2{c, Fp::zero()};
        else *out = Fp2{Fp::zero(), c};
        return out->sqr() == a;
    }
    const Fp n = a.a0.sqr() + a.a1.sqr();
    const Fp s = fp_pow_qm3_4(n, true);
    if (s.sqr() != n) return false;  // the norm is not a square: neither is a
    Fp t = (a.a0 + s) * half;
    // (t = 0 would need a0 = -s, i.e. a1^2 = s^2 - a0^2 = 0: not on this branch)
    const Fp e = fp_pow_qm3_4(t, false), c = e * t;
    const Fp w = a.a1 * (c * e.sqr() * half);  // a1 / (2 c)
    if (c.sqr() == t) *out = Fp2{c, w};
    else *out = Fp2{w, c};
    return out->sqr() == a;
}
__device__ __forceinline__ bool f2_lex_largest_dev(const Fp2& y) {  // gnark-crypto: compares A1 first, A0 when A1 = 0
    return y.a1.is_zero() ? fp_lex_largest_dev(y.a0.from_mont()) : fp_lex_largest_dev(y.a1.from_mont());
}
// G2Affine.SetBytes on a compressed encoding (X.A1 | X.A0 big-endian, flags on the first byte) with the subgroup check the gnark-crypto Decoder
// applies by default (r * P = infinity: the twist has a cofactor).  bt = 3 / (9 + u), Montgomery.
// r-torsion membership on the twist, with the untwist-Frobenius-twist endomorphism psi: (x, y) -> (conj(x) * gx, conj(y) * gy), gx = xi^((q-1)/3),
// gy = xi^((q-1)/2), xi = 9 + u, which acts on G2 as multiplication by q = 6 x0^2 (mod r); x0 = 4965661367192848881.  Two exact tests (both accept exactly
// the points r * P = infinity accepts; tests/test_gpu_keyio.py holds twist points outside G2 and a G2 point shifted by a cofactor-torsion point):
//   psi(P) == [6 x0^2] P                                       127 doublings + 64 additions     (rounds 2-4)
//   [x0 + 1] P + psi([x0] P) + psi^2([x0] P) == psi^3([2 x0] P)  63 doublings + 27 + 4 additions  (eprint 2022/348 sec. 5.1 for BN curves; gnark-crypto's
//                                                               G2Jac.IsInSubGroup): ONE multiplication by the 63-bit x0, three psi, a few additions -- half
//                                                               the work of a decompression's larger half (68 -> see DESIGN.md 3.8 per 2^20 points)
struct PsiConsts { Fp2 gx, gy; Fp half; };  // psi's two coefficients; 1 / 2 for the square root
// The second test runs on the 29-bit multiplier (ff29.hpp: acc29g2_dbl / acc29g2_add, whose class invariant -- every coordinate component < 32 p, weakly
// normalised, in and out -- tools/u29_model.py proves): 63 doublings and 27 + 3 full additions with one reduction per output component instead of three saturated
// products per Fp2 product (29.5-30.5 ms per 2^20 points against 33.4-34.2 for the saturated form in the same kernel: profiles/rnd5_v_g2_subgroup_variants.txt).
// psi keeps the invariant: X and Y times a contracted constant come out below 1.5 p; the conjugated ZZ / ZZZ components are contracted.
__device__ __forceinline__ Acc29G2 g2_psi_dev29(const Acc29G2& t, const U29x2& gx, const U29x2& gy) {
    if (t.inf) return t;
    const U29 one = u29_one();
    Acc29G2 r;
    r.inf = false;
    // conj(v) * g = (v0 g0 + v1 g1) + (v0 g1 - v1 g0) u
    r.x = U29x2{u29_mul2(t.x.c0, gx.c0, t.x.c1, gx.c1), u29_mul2(t.x.c0, gx.c1, u29_neg<32>(t.x.c1), gx.c0)};
    r.y = U29x2{u29_mul2(t.y.c0, gy.c0, t.y.c1, gy.c1), u29_mul2(t.y.c0, gy.c1, u29_neg<32>(t.y.c1), gy.c0)};
    r.zz = U29x2{t.zz.c0, u29_mul(u29_neg<32>(t.zz.c1), one)};
    r.zzz = U29x2{t.zzz.c0, u29_mul(u29_neg<32>(t.zzz.c1), one)};
    return r;
}
__device__ __forceinline__ bool f2_eq29(const U29x2& a, const U29x2& b) {  // exact: through the canonical images
    const Fp2 x = f2_store29(a), y = f2_store29(b);
    return x == y;
}
// *pp is read again wherever P is added (28 times: 128 bytes from L2) instead of being held in 72 registers next to the accumulator and an addition's temporaries
__device__ __forceinline__ void g2_add_affine29(Acc29G2& a, const Affine<Fp2>* __restrict__ pp) {
    const Fp2 one2{Fp::one(), Fp::zero()};
    Acc29G2 P1;
    const Affine<Fp2> q = *pp;
    acc29g2_load(P1, XYZZ<Fp2>{q.x, q.y, one2, one2});
    acc29g2_add(a, P1);
}
// [x0] P, the long half of the test (63 doublings, 27 additions): a kernel of its own under a two-waves-per-SIMD register bound (256 registers and 132 bytes of
// scratch per lane; with the tail in the same kernel: 256 + 140 registers, one wave per SIMD)
__global__ __launch_bounds__(128, 2) void k_g2_x0_mul(const Affine<Fp2>* __restrict__ pts, size_t n, XYZZ<Fp2>* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (pts[i].is_inf()) return;
    const uint32_t x0[2] = {0x4a6909f1u, 0x44e992b4u};  // 4965661367192848881
    Acc29G2 a;
    a.inf = true;
    a.x = a.y = a.zz = a.zzz = f2_load29(pts[i].x);  // (defined values; never read while inf)
#pragma unroll 1
    for (int k = 62; k >= 0; k--) {
        acc29g2_dbl(a);
        if ((x0[k >> 5] >> (k & 31)) & 1) g2_add_affine29(a, pts + i);
    }
    out[i] = acc29g2_to_xyzz(a);
}
// the rest: [x0 + 1] P + psi([x0] P) + psi^2([x0] P) == psi^3([2 x0] P)
__device__ bool g2_subgroup_tail29(const Affine<Fp2>* __restrict__ pp, const XYZZ<Fp2>& x0p, const PsiConsts& K) {
    Acc29G2 a;
    acc29g2_load(a, x0p);
    const U29x2 gx = f2_contract29(f2_load29(K.gx)), gy = f2_contract29(f2_load29(K.gy));
    const Acc29G2 b = g2_psi_dev29(a, gx, gy);  // psi([x0] P)
    g2_add_affine29(a, pp);                      // [x0 + 1] P
    Acc29G2 lhs = a;
    acc29g2_add(lhs, b);
    const Acc29G2 c = g2_psi_dev29(b, gx, gy);  // psi^2([x0] P)
    acc29g2_add(lhs, c);
    Acc29G2 d = g2_psi_dev29(c, gx, gy);        // psi^3([x0] P)
    acc29g2_dbl(d);                              // psi^3([2 x0] P)
    if (lhs.inf || d.inf) return lhs.inf && d.inf;
    // (a coordinate sum that came out as the point at infinity went through the canonical path of acc29g2_add / _dbl, which sets .inf)
    return f2_eq29(f2_mulFK29<40>(lhs.x, d.zz), f2_mulFK29<40>(d.x, lhs.zz)) && f2_eq29(f2_mulFK29<40>(lhs.y, d.zzz), f2_mulFK29<40>(d.y, lhs.zzz));
}
__global__ __launch_bounds__(128) void k_g2_decompress(const uint32_t* __restrict__ raw, size_t n, Fp2 bt, PsiConsts psi, Affine<Fp2>* __restrict__ out,
                                                       int* __restrict__ status) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp x1 = load_be32<Fp>(raw + 16 * i), x0 = load_be32<Fp>(raw + 16 * i + 8);
    const uint32_t flag = x1.l[7] >> 30;
    x1.l[7] &= 0x3fffffffu;
    Affine<Fp2> p = Affine<Fp2>::inf();
    if (flag == 1) {
        if (!x1.is_zero() || !x0.is_zero()) atomicOr(status, 8);
        out[i] = p;
        return;
    }
    if (flag == 0 || geq_mod<FpParams>(x1.l) || geq_mod<FpParams>(x0.l)) {
        atomicOr(status, 8);
        out[i] = p;
        return;
    }
    Fp2 x{x0.to_mont(), x1.to_mont()};
    Fp2 rhs = x.sqr() * x + bt, y;
    if (!f2_sqrt_dev(rhs, psi.half, &y)) {
        atomicOr(status, 8);
        out[i] = p;
        return;
    }
    if (f2_lex_largest_dev(y) != (flag == 3)) y = y.neg();
    p.x = x;
    p.y = y;
    out[i] = p;
}
// second half of G2Affine.SetBytes, a kernel of its own (the square root and the subgroup test in one kernel need 512 registers: one wave per SIMD; apart, each
// runs with two or more): a point outside the r-torsion subgroup becomes the point at infinity and sets status bit 16
template <bool FULL>
__global__ __launch_bounds__(128) void k_g2_subgroup(Affine<Fp2>* __restrict__ pts, size_t n, const XYZZ<Fp2>* __restrict__ x0p, PsiConsts psi, int* __restrict__ status) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (pts[i].is_inf()) return;
    bool member;
    if (FULL) {  // ZKMI_G2_FULL_SUBGROUP_CHECK=1: the definition, r * P == infinity (A/B switch, four times the work)
        const uint32_t rk[8] = {FrParams::MOD[0], FrParams::MOD[1], FrParams::MOD[2], FrParams::MOD[3], FrParams::MOD[4], FrParams::MOD[5], FrParams::MOD[6], FrParams::MOD[7]};
        const Affine<Fp2> p = pts[i];
        member = scalar_mul(p, rk).is_inf();
    } else {
        member = g2_subgroup_tail29(pts + i, x0p[i], psi);
    }
    if (!member) {
        atomicOr(status, 16);
        pts[i] = Affine<Fp2>::inf();
    }
}
// idx == nullptr: point i; else point idx[i] (a key is stored wire-indexed and written without its points at infinity)
__global__ void k_g2_compress(const Affine<Fp2>* __restrict__ pts, const uint32_t* __restrict__ idx, size_t n, uint32_t* __restrict__ raw) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<Fp2> p = pts[idx ? idx[i] : i];
    Fp x1 = Fp::zero(), x0 = Fp::zero();
    uint32_t flag = 1;
    if (!p.is_inf()) {
        x1 = p.x.a1.from_mont();
        x0 = p.x.a0.from_mont();
        flag = f2_lex_largest_dev(p.y) ? 3 : 2;
    }
    x1.l[7] |= flag << 30;
    store_be32(raw + 16 * i, x1);
    store_be32(raw + 16 * i + 8, x0);
}
__global__ void k_g1_compress_idx(const Affine<Fp>* __restrict__ pts, const uint32_t* __restrict__ idx, size_t n, uint32_t* __restrict__ raw) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<Fp> p = pts[idx[i]];
    Fp x = Fp::zero();
    uint32_t flag = 1;
    if (!p.is_inf()) {
        x = p.x.from_mont();
        flag = fp_lex_largest_dev(p.y.from_mont()) ? 3 : 2;
    }
    x.l[7] |= flag << 30;
    store_be32(raw + 8 * i, x);
}
// InfinityA / InfinityB of a wire-indexed array: one byte per point
template <class F>
__global__ void k_inf_flags(const Affine<F>* __restrict__ pts, size_t n, uint8_t* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = pts[i].is_inf() ? 1 : 0;
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

static HFp2 f2_pow(HFp2 a, const uint64_t e[4]);
int g2_decompress_dev(Slot* s, hipStream_t st, const void* d_raw, size_t n, void* d_out, int* d_status) {
    HFp nine = HFp::zero(), three = HFp::one() + HFp::one() + HFp::one();
    for (int i = 0; i < 3; i++) nine = nine + three;
    const HFp2 bt = HFp2{three, HFp::zero()} * HFp2{nine, HFp::one()}.inv();
    Fp2 btd;
    memcpy(&btd, &bt, sizeof btd);
    // psi's coefficients: xi^((q-1)/3), xi^((q-1)/2)
    static const uint64_t E3[4] = {0x69602eb24829a9c2ULL, 0xdd2b2385cd7b4384ULL, 0xe81ac1e7808072c9ULL, 0x10216f7ba065e00dULL};
    static const uint64_t E2h[4] = {0x9e10460b6c3e7ea3ULL, 0xcbc0b548b438e546ULL, 0xdc2822db40c0ac2eULL, 0x183227397098d014ULL};
    const HFp2 xi{nine, HFp::one()}, gx = f2_pow(xi, E3), gy = f2_pow(xi, E2h);
    PsiConsts psi;
    memcpy(&psi.gx, &gx, sizeof gx);
    memcpy(&psi.gy, &gy, sizeof gy);
    const HFp half = (HFp::one() + HFp::one()).inv();
    memcpy(&psi.half, &half, sizeof half);
    static const int full = ZK_EXP("ZKMI_G2_FULL_SUBGROUP_CHECK", 0);
    if (n) {
        ZK_LAUNCH(s, st, "g2_decompress", k_g2_decompress, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, (const uint32_t*)d_raw, n, btd, psi, (Affine<Fp2>*)d_out, d_status);
        const dim3 grid((unsigned)((n + 127) / 128));
        if (full) {
            ZK_LAUNCH(s, st, "g2_subgroup", (k_g2_subgroup<true>), grid, dim3(128), 0, (Affine<Fp2>*)d_out, n, (const XYZZ<Fp2>*)nullptr, psi, d_status);
        } else {
            XYZZ<Fp2>* x0p = (XYZZ<Fp2>*)s->alloc(n * sizeof(XYZZ<Fp2>));  // [x0] P of every point, between the two kernels of the test (the caller reserved it: G2_DECOMPRESS_SCRATCH)
            if (!x0p) return set_err(ZK_ERR_ARG, "G2 decompression: the slot's workspace has no room for %zu scratch points", n);
            ZK_LAUNCH(s, st, "g2_x0_mul", k_g2_x0_mul, grid, dim3(128), 0, (const Affine<Fp2>*)d_out, n, x0p);
            ZK_LAUNCH(s, st, "g2_subgroup", (k_g2_subgroup<false>), grid, dim3(128), 0, (Affine<Fp2>*)d_out, n, (const XYZZ<Fp2>*)x0p, psi, d_status);
        }
    }
    return ZK_OK;
}
int g2_compress_dev(Slot* s, hipStream_t st, const void* d_pts, const uint32_t* d_idx, size_t n, void* d_raw) {
    if (n) ZK_LAUNCH(s, st, "g2_compress", k_g2_compress, dim3(grid1(n)), dim3(256), 0, (const Affine<Fp2>*)d_pts, d_idx, n, (uint32_t*)d_raw);
    return ZK_OK;
}
int g1_compress_idx_dev(Slot* s, hipStream_t st, const void* d_pts, const uint32_t* d_idx, size_t n, void* d_raw) {
    if (n) ZK_LAUNCH(s, st, "g1_compress", k_g1_compress_idx, dim3(grid1(n)), dim3(256), 0, (const Affine<Fp>*)d_pts, d_idx, n, (uint32_t*)d_raw);
    return ZK_OK;
}
int inf_flags_dev(Slot* s, hipStream_t st, int is_g2, const void* d_pts, size_t n, void* d_out) {
    if (!n) return ZK_OK;
    if (is_g2) ZK_LAUNCH(s, st, "inf_flags", k_inf_flags<Fp2>, dim3(grid1(n)), dim3(256), 0, (const Affine<Fp2>*)d_pts, n, (uint8_t*)d_out);
    else ZK_LAUNCH(s, st, "inf_flags", k_inf_flags<Fp>, dim3(grid1(n)), dim3(256), 0, (const Affine<Fp>*)d_pts, n, (uint8_t*)d_out);
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
    if (flag == 1) {  // infinity: canonical only -- every other bit zero, as g1_decompress_host and gnark-crypto's SetBytes require
        if (in[0] & 0x3f) return false;
        for (int i = 1; i < 64; i++)
            if (in[i]) return false;
        return true;
    }
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
    SrsHeader H;  // the count and the total length are settled on the host, from the caller's bytes alone (text_host.hpp), before anything is sized by them
    {
        std::string e;
        const int rc = kzg_srs_header(data, len, is_hex, &H, &e);
        if (rc != ZK_OK) return set_err(rc, "%s", e.c_str());
    }
    const size_t nbytes = H.nbytes, n = H.n_g1;
    const uint8_t* head = H.g2;
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
        ZK_TRY(h2d_big(d_text, data, len, st));
        ZK_TRY(hex_decode_dev(s, st, d_text, nbytes, d_bytes, d_status));
    } else {
        ZK_TRY(h2d_big(d_bytes, data, nbytes, st));
    }
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

// The two G2 points of an SRS image and nothing else: what plonk.Verify takes from the SRS (kzg.Verify's pairing check).  Host only -- no device is
// touched, so a process that only verifies (nargo verify) never starts the HIP runtime.  The header and the length are checked as in _read; the G1 points are not looked at.
int zk_bn254_kzg_srs_g2(const void* data, size_t len, int is_hex, zk_g2_affine g2_out[2]) {
    if (!data || !g2_out) return set_err(ZK_ERR_ARG, "null pointer");
    SrsHeader H;
    {
        std::string e;
        const int rc = kzg_srs_header(data, len, is_hex, &H, &e);
        if (rc != ZK_OK) return set_err(rc, "%s", e.c_str());
    }
    Affine<HFp2> g2[2];
    for (int k = 0; k < 2; k++)
        if (!g2_decompress_host(H.g2 + 64 * k, &g2[k])) return set_err(ZK_ERR_ARG, "SRS: invalid G2 point %d", k);
    memcpy(g2_out, g2, 256);
    return ZK_OK;
}

// kzg.SRS.WriteTo of a registered G1 base array + the two G2 points: bytes (or hex text) into out; *out_len = bytes written.
int zk_bn254_kzg_srs_write(uint64_t handle, const zk_g2_affine g2[2], int as_hex, void* out, size_t cap, size_t* out_len) {
    ZK_ON_ENTRY_OF(handle);
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

// ---- Groth16 keys (gnark v0.8.0 internal/backend/bn254/groth16/marshal.go  [UPSTREAM-RECALL]) -- what the reference's intended Groth16 FFI moves as hex:
// ProveWithPK(rawR1CS, encodedProvingKey) -> provingKey.ReadFrom  [REF gnark_backend_ffi/backend/groth16/r1cs.go:107-143], Preprocess -> both keys
// out [REF r1cs.go:214-266].  ProvingKey.WriteTo (compressed):
//   Domain.WriteTo 168 B | G1.Alpha, G1.Beta, G1.Delta 3 x 32 | G1.A, G1.B, G1.Z, G1.K each u32 count + 32 B per point | G2.Beta, G2.Delta 2 x 64
//   | G2.B u32 count + 64 B per point | nbWires u64 | NbInfinityA u64 | NbInfinityB u64 | InfinityA, InfinityB: nbWires bytes (0 / 1) each, no prefix
// A, B, G2.B are stored WITHOUT their points at infinity (that is what the bitmaps are for); the public-wire count is nbWires - len(K).
namespace {
void put_be(uint8_t* o, uint64_t v, int n) { for (int i = 0; i < n; i++) o[i] = (uint8_t)(v >> (8 * (n - 1 - i))); }
void domain_bytes(const Domain* d, uint8_t o[168]) {
    put_be(o, (uint64_t)1 << d->logn, 8);
    fr_to_be(d->card_inv, o + 8);
    fr_to_be(d->gen, o + 40);
    fr_to_be(d->gen_inv, o + 72);
    fr_to_be(d->coset, o + 104);
    fr_to_be(d->coset_inv, o + 136);
}
struct DevFree {
    std::vector<void*> ptrs;
    bool keep = false;
    int alloc(void** d, size_t bytes) {
        ZK_HIP(hipMalloc(d, bytes ? bytes : 16));
        ptrs.push_back(*d);
        return ZK_OK;
    }
    ~DevFree() { if (!keep) for (void* q : ptrs) (void)hipFree(q); }
};
}  // namespace

// groth16.ProvingKey.ReadFrom on the bytes (or hex text) of ProvingKey.WriteTo: every point is decompressed on the device (G1: one square root; G2:
// an Fp2 square root and the subgroup check), A / B / G2.B are expanded to the wire-indexed resident layout and the window tables are built as in
// zk_bn254_groth16_pk_load.  flags: bit 0 of zk_groth16_pk.flags (no window tables); table_window_bits as there.
int zk_bn254_groth16_pk_read(const void* data, size_t len, int is_hex, int flags, int table_window_bits, uint64_t* handle) {
    if (!data || !handle) return set_err(ZK_ERR_ARG, "null pointer");
    if (flags & ~1) return set_err(ZK_ERR_ARG, "only flag bit 0 (no window tables) applies to a key read from its wire format");
    struct Lap {
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        void lap(const char* name) {
            const auto t1 = std::chrono::steady_clock::now();
            prof_host(name, std::chrono::duration<double, std::milli>(t1 - t0).count());
            t0 = t1;
        }
    } lap;
    Groth16KeyHeader H;  // the section table, the counts and the two bitmaps: read and cross-checked on the host from the caller's bytes alone (text_host.hpp)
    {
        std::string e;
        const int rc = groth16_pk_header(data, len, is_hex, &H, &e);
        if (rc != ZK_OK) return set_err(rc, "%s", e.c_str());
    }
    lap.lap("export.pk_read_header_host");
    const unsigned logN = H.logN;
    const size_t* cnt = H.cnt;
    const size_t* at = H.at;
    const size_t at_g2 = H.at_g2, at_bm = H.at_bitmaps;
    const uint64_t nw = H.n_wires, nia = H.nb_inf_a, nib = H.nb_inf_b;
    const std::vector<uint8_t>&ia = H.inf_a, &ib = H.inf_b;
    const uint8_t* dom_in = H.domain;
    uint8_t dom_ok[168];

    // two slots: the first one's stream carries the uploads (and its arena the text), the second one's the decompression -- two ORDINARY streams, because a lean
    // process (the export shim's) has no high-priority stream yet and hi() would hand back the slot's own: upload and decompression would take turns
    SlotsGuard<2> g;
    ZK_TRY(acquire_slots(2, g.s));
    Slot* s = g.s[0];
    hipStream_t st = s->stream;
    Domain* dom;
    ZK_TRY(get_domain(s, st, logN, 0, &dom));
    domain_bytes(dom, dom_ok);
    if (memcmp(dom_in, dom_ok, 168)) return set_err(ZK_ERR_ARG, "proving key: the domain is not gnark-crypto's radix-2 domain of that size");
    lap.lap("export.pk_read_header_domain");
    // Everything up to the bitmaps goes to the device (every section starts at a multiple of 4 bytes by construction) -- G2.B FIRST: its points are the expensive
    // ones to decompress (an Fp2 square root and the subgroup test each), so their kernel runs on the slot's second stream while the other 0.24 GB of a
    // 2^20-constraint key's text are still crossing PCIe on the first (round 5: 41 + 17 + 68 ms one after the other before).
    lap.lap("export.pk_read_slot");
    ZK_TRY(s->reserve((is_hex ? 2 * at_bm : 0) + at_bm + (cnt[4] + 2) * G2_DECOMPRESS_SCRATCH + 8 * 4096));
    lap.lap("export.pk_read_reserve");
    int* d_status = (int*)s->alloc(64);
    uint8_t* d_bytes = (uint8_t*)s->alloc(at_bm + 16);
    uint8_t* d_text = is_hex ? (uint8_t*)s->alloc(2 * at_bm + 16) : nullptr;
    hipStream_t sx = g.s[1]->stream;  // decompression
    ZK_HIP(hipMemsetAsync(d_status, 0, 4, st));
    auto upload = [&](size_t from, size_t to) -> int {  // bytes [from, to) of the key image -> d_bytes, on `st`
        if (to <= from) return ZK_OK;
        if (is_hex) {
            ZK_TRY(h2d_big(d_text + 2 * from, (const char*)data + 2 * from, 2 * (to - from), st));
            ZK_TRY(hex_decode_dev(s, st, d_text + 2 * from, to - from, d_bytes + from, d_status));
        } else {
            ZK_TRY(h2d_big(d_bytes + from, (const char*)data + from, to - from, st));
        }
        return ZK_OK;
    };
    hipEvent_t ev_g2 = nullptr, ev_rest = nullptr;
    struct Ev { hipEvent_t* e[2]; ~Ev() { for (auto p : e) if (*p) (void)hipEventDestroy(*p); } } ev_guard{{&ev_g2, &ev_rest}};
    ZK_HIP(hipEventCreateWithFlags(&ev_g2, hipEventDisableTiming));
    ZK_HIP(hipEventCreateWithFlags(&ev_rest, hipEventDisableTiming));
    DevFree tmp, own;  // tmp: the compact A / B / G2.B and the five single points; own: K and Z, adopted by the key
    void *d_a = nullptr, *d_b = nullptr, *d_b2 = nullptr, *d_k = nullptr, *d_z = nullptr, *d_single = nullptr;
    ZK_TRY(tmp.alloc(&d_a, cnt[0] * 64));
    ZK_TRY(tmp.alloc(&d_b, cnt[1] * 64));
    ZK_TRY(tmp.alloc(&d_b2, cnt[4] * 128));
    ZK_TRY(tmp.alloc(&d_single, 3 * 64 + 2 * 128));
    ZK_TRY(own.alloc(&d_z, cnt[2] * 64));
    ZK_TRY(own.alloc(&d_k, cnt[3] * 64));
    const size_t g2_end = at[4] + cnt[4] * 64;
    lap.lap("export.pk_read_allocs");
    {   // in quarters: the first points are being decompressed while the last ones are still on their way (an event may be re-recorded once the wait on its
        // previous recording has been enqueued)
        const size_t parts = cnt[4] >= ((size_t)1 << 16) ? 4 : 1, per = (cnt[4] + parts - 1) / parts;
        for (size_t q = 0; q < parts; q++) {
            const size_t p0 = q * per, p1 = p0 + per < cnt[4] ? p0 + per : cnt[4];
            if (p1 <= p0) break;
            ZK_TRY(upload(at[4] + p0 * 64, at[4] + p1 * 64));
            ZK_HIP(hipEventRecord(ev_g2, st));
            ZK_HIP(hipStreamWaitEvent(sx, ev_g2, 0));
            ZK_TRY(g2_decompress_dev(s, sx, d_bytes + at[4] + p0 * 64, p1 - p0, (uint8_t*)d_b2 + p0 * 128, d_status));
        }
    }
    lap.lap("export.pk_read_upload_g2_part");
    ZK_TRY(upload(0, at[4]));
    ZK_TRY(upload(g2_end, at_bm));
    ZK_HIP(hipEventRecord(ev_rest, st));
    ZK_HIP(hipStreamWaitEvent(sx, ev_rest, 0));
    lap.lap("export.pk_read_upload_rest");
    ZK_TRY(g1_decompress_dev(s, sx, d_bytes + 168, 3, d_single, d_status));
    ZK_TRY(g2_decompress_dev(s, sx, d_bytes + at_g2, 2, (uint8_t*)d_single + 192, d_status));
    void* g1_dst[4] = {d_a, d_b, d_z, d_k};
    for (int k = 0; k < 4; k++) ZK_TRY(g1_decompress_dev(s, sx, d_bytes + at[k], cnt[k], g1_dst[k], d_status));
    st = sx;  // the results are read back behind the decompression
    uint8_t single[3 * 64 + 2 * 128];
    int h_status = 0;
    ZK_HIP(hipMemcpyAsync(single, d_single, sizeof single, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipMemcpyAsync(&h_status, d_status, 4, hipMemcpyDeviceToHost, st));
    ZK_TRY(slot_sync(s, st));
    lap.lap("export.pk_read_decompress_wait");
    if (h_status & 1) return set_err(ZK_ERR_ARG, "proving key: invalid hex character");
    if (h_status & 4) return set_err(ZK_ERR_ARG, "proving key: invalid compressed G1 point (bad flags, x >= q, or no square root)");
    if (h_status & 8) return set_err(ZK_ERR_ARG, "proving key: invalid compressed G2 point (bad flags, coordinate >= q, or no square root)");
    if (h_status & 16) return set_err(ZK_ERR_ARG, "proving key: a G2 point outside the r-torsion subgroup");
    s->reset();
    for (int i = 0; i < 2; i++) {  // pk_load takes slots of its own
        release_slot(g.s[i]);
        g.s[i] = nullptr;
    }
    zk_groth16_pk pk;
    memset(&pk, 0, sizeof pk);
    pk.log_domain = logN;
    pk.n_wires = nw;
    pk.n_public = nw - cnt[3];
    pk.g1_alpha = (const zk_g1_affine*)single;
    pk.g1_beta = (const zk_g1_affine*)(single + 64);
    pk.g1_delta = (const zk_g1_affine*)(single + 128);
    pk.g2_beta = (const zk_g2_affine*)(single + 192);
    pk.g2_delta = (const zk_g2_affine*)(single + 320);
    pk.g1_a = (const zk_g1_affine*)d_a;
    pk.g1_b = (const zk_g1_affine*)d_b;
    pk.g1_k = (const zk_g1_affine*)d_k;
    pk.g1_z = (const zk_g1_affine*)d_z;
    pk.g2_b = (const zk_g2_affine*)d_b2;
    pk.bases_on_device = 1;
    pk.flags = flags;
    pk.infinity_a = ia.data();
    pk.infinity_b = ib.data();
    pk.nb_infinity_a = nia;
    pk.nb_infinity_b = nib;
    pk.table_window_bits = table_window_bits;
    ZK_TRY(zk_bn254_groth16_pk_load(&pk, handle));
    lap.lap("export.pk_read_load");
    ZK_TRY(groth16_pk_adopt(*handle));  // K and Z were allocated here for the key; A / B / G2.B are its own expanded arrays already
    own.keep = true;
    return ZK_OK;
}

// groth16.ProvingKey.WriteTo of a resident key: bytes (or hex text) into out; *out_len = bytes needed / written (out == NULL: size query).
int zk_bn254_groth16_pk_write(uint64_t handle, int as_hex, void* out, size_t cap, size_t* out_len) {
    ZK_ON_ENTRY_OF(handle);
    if (!out_len) return set_err(ZK_ERR_ARG, "null pointer");
    Groth16View v;
    ZK_TRY(groth16_pk_view(handle, &v));
    const size_t N = (size_t)1 << v.log_domain, nw = v.n_wires, nk = nw - v.n_public;
    if (v.nz != N - 1 && N > 1) return set_err(ZK_ERR_ARG, "a range-sharded slice of a key has no gnark wire format");
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    Slot* s = g.s;
    hipStream_t st = s->stream;
    // InfinityA / InfinityB from the wire-indexed arrays
    ZK_TRY(s->reserve(2 * nw + 4096));
    uint8_t* d_flags = (uint8_t*)s->alloc(2 * nw + 16);
    std::vector<uint8_t> ia(nw ? nw : 1), ib(nw ? nw : 1);
    ZK_TRY(inf_flags_dev(s, st, 0, v.d_a, nw, d_flags));
    ZK_TRY(inf_flags_dev(s, st, 0, v.d_b, nw, d_flags + nw));
    if (nw) {
        ZK_HIP(hipMemcpyAsync(ia.data(), d_flags, nw, hipMemcpyDeviceToHost, st));
        ZK_HIP(hipMemcpyAsync(ib.data(), d_flags + nw, nw, hipMemcpyDeviceToHost, st));
    }
    ZK_TRY(slot_sync(s, st));
    std::vector<uint32_t> idx_a, idx_b;
    for (size_t i = 0; i < nw; i++) {
        if (!ia[i]) idx_a.push_back((uint32_t)i);
        if (!ib[i]) idx_b.push_back((uint32_t)i);
    }
    const size_t na = idx_a.size(), nb = idx_b.size();
    const size_t at_a = 168 + 96 + 4, at_b = at_a + 32 * na + 4, at_z = at_b + 32 * nb + 4, at_k = at_z + 32 * N + 4, at_g2 = at_k + 32 * nk, at_b2 = at_g2 + 128 + 4,
                 at_cnt = at_b2 + 64 * nb, at_bm = at_cnt + 24, nbytes = at_bm + 2 * nw, need = as_hex ? 2 * nbytes : nbytes;
    *out_len = need;
    if (!out) return ZK_OK;
    if (cap < need) return set_err(ZK_ERR_ARG, "output holds %zu bytes, %zu needed", cap, need);
    s->reset();
    const size_t padded = align_up(nbytes, 4);
    ZK_TRY(s->reserve(3 * padded + 4 * (na + nb) + 65536));
    uint8_t* d_bytes = (uint8_t*)s->alloc(padded + 16);
    uint32_t* d_ia = (uint32_t*)s->alloc(4 * na + 16);
    uint32_t* d_ib = (uint32_t*)s->alloc(4 * nb + 16);
    if (na) ZK_HIP(hipMemcpyAsync(d_ia, idx_a.data(), 4 * na, hipMemcpyHostToDevice, st));
    if (nb) ZK_HIP(hipMemcpyAsync(d_ib, idx_b.data(), 4 * nb, hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemsetAsync(d_bytes + (padded - 4), 0, 4, st));
    // host-built pieces: the head (domain, three G1 points, count of A), the counts, the two G2 points, the trailer
    Domain* dom;
    ZK_TRY(get_domain(s, st, v.log_domain, 0, &dom));
    std::vector<uint8_t> head(at_a), mid(128 + 4), trailer(24 + 2 * nw);
    domain_bytes(dom, head.data());
    g1_compress(v.alpha, head.data() + 168);
    g1_compress(v.beta, head.data() + 200);
    g1_compress(v.delta, head.data() + 232);
    put_be(head.data() + 264, na, 4);
    uint8_t c_b[4], c_z[4], c_k[4];
    put_be(c_b, nb, 4);
    put_be(c_z, N, 4);
    put_be(c_k, nk, 4);
    g2_compress(v.beta2, mid.data());
    g2_compress(v.delta2, mid.data() + 64);
    put_be(mid.data() + 128, nb, 4);
    put_be(trailer.data(), nw, 8);
    put_be(trailer.data() + 8, nw - na, 8);
    put_be(trailer.data() + 16, nw - nb, 8);
    if (nw) {
        memcpy(trailer.data() + 24, ia.data(), nw);
        memcpy(trailer.data() + 24 + nw, ib.data(), nw);
    }
    ZK_HIP(hipMemcpyAsync(d_bytes, head.data(), at_a, hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(d_bytes + at_b - 4, c_b, 4, hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(d_bytes + at_z - 4, c_z, 4, hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(d_bytes + at_k - 4, c_k, 4, hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(d_bytes + at_g2, mid.data(), 132, hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(d_bytes + at_cnt, trailer.data(), trailer.size(), hipMemcpyHostToDevice, st));
    ZK_TRY(g1_compress_idx_dev(s, st, v.d_a, d_ia, na, d_bytes + at_a));
    ZK_TRY(g1_compress_idx_dev(s, st, v.d_b, d_ib, nb, d_bytes + at_b));
    ZK_TRY(g1_compress_dev(s, st, v.d_z, N, d_bytes + at_z));
    ZK_TRY(g1_compress_dev(s, st, v.d_k, nk, d_bytes + at_k));
    ZK_TRY(g2_compress_dev(s, st, v.d_b2, d_ib, nb, d_bytes + at_b2));
    if (as_hex) {
        void* d_text = s->alloc(2 * padded + 16);
        ZK_TRY(hex_encode_dev(s, st, d_bytes, padded, d_text));
        ZK_HIP(hipMemcpyAsync(out, d_text, 2 * nbytes, hipMemcpyDeviceToHost, st));
    } else {
        ZK_HIP(hipMemcpyAsync(out, d_bytes, nbytes, hipMemcpyDeviceToHost, st));
    }
    return slot_sync(s, st);  // the host staging vectors live until here
}

// groth16.VerifyingKey.WriteTo (compressed): [alpha]1, [beta]1, [beta]2, [gamma]2, [delta]1, [delta]2, u32 len(K), K  -- from what
// zk_bn254_groth16_setup returns (vk_g1 = [alpha]1 then the n_k = n_public points K; vk_g2 = [beta]2, [gamma]2, [delta]2) and the resident
// proving key ([beta]1, [delta]1).  Host only: 292 + 32 * n_k bytes.
int zk_bn254_groth16_vk_write(uint64_t pk_handle, const zk_g1_affine* vk_g1, size_t n_k, const zk_g2_affine vk_g2[3], int as_hex, void* out, size_t cap,
                              size_t* out_len) {
    ZK_ON_ENTRY_OF(pk_handle);
    if (!vk_g1 || !vk_g2 || !out_len) return set_err(ZK_ERR_ARG, "null pointer");
    Groth16View v;
    ZK_TRY(groth16_pk_view(pk_handle, &v));
    if (n_k != v.n_public) return set_err(ZK_ERR_LEN, "verifying key: %zu K points, the key has %zu public wires", n_k, v.n_public);
    const size_t nbytes = 292 + 32 * n_k, need = as_hex ? 2 * nbytes : nbytes;
    *out_len = need;
    if (!out) return ZK_OK;
    if (cap < need) return set_err(ZK_ERR_ARG, "output holds %zu bytes, %zu needed", cap, need);
    std::vector<uint8_t> b(nbytes);
    const Affine<HFp>* g1 = (const Affine<HFp>*)vk_g1;
    const Affine<HFp2>* g2 = (const Affine<HFp2>*)vk_g2;
    g1_compress(g1[0], b.data());
    g1_compress(v.beta, b.data() + 32);
    g2_compress(g2[0], b.data() + 64);
    g2_compress(g2[1], b.data() + 128);
    g1_compress(v.delta, b.data() + 192);
    g2_compress(g2[2], b.data() + 224);
    put_be(b.data() + 288, n_k, 4);
    for (size_t i = 0; i < n_k; i++) g1_compress(g1[1 + i], b.data() + 292 + 32 * i);
    if (!as_hex) { memcpy(out, b.data(), nbytes); return ZK_OK; }
    static const char* dg = "0123456789abcdef";
    char* o = (char*)out;
    for (size_t i = 0; i < nbytes; i++) { o[2 * i] = dg[b[i] >> 4]; o[2 * i + 1] = dg[b[i] & 15]; }
    return ZK_OK;
}


}  // extern "C"
