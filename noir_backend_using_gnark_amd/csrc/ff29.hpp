// Unsaturated Fp arithmetic for the MSM accumulate kernels on gfx950: 9 limbs of 29 bits in 32-bit registers,
// Montgomery radix R' = 2^261, lazily reduced values.
//
// Same mathematical object as ff.hpp's Field<FpParams> (gnark-crypto `fp.Element`, /root/reference/gnark_backend_ffi/go.mod:5),
// different in-register representation.  Memory keeps gnark's image (8 x u32, Montgomery 2^256, canonical):
//     load : canonical V = x*2^256 mod p  ->  X = V << 5   (= x*2^261 mod p up to a multiple of p; X < 32 p)
//     store: Montgomery-multiply by 2^256, ripple-normalise, subtract 2p / p  ->  canonical V again
// Why: measured on MI355X (tools/ubench*.hip) a v_mad_u64_u32 costs the same issue time (~4.6 cycles/wave) as the carry
// instruction that must follow it in the saturated schedule.  With 29-bit limbs a 64-bit column accumulator takes all 18
// products of a column with no carry handling: 206 instructions per product instead of ~305; additions are 9 plain
// v_add_u32; subtractions add a multiple of p whose limbs dominate the subtrahend's ("bias") and re-normalise in one
// parallel pass.  Values are NOT canonical in flight: tools/u29_model.py propagates worst-case bounds through the exact
// operation sequence of xyzz_madd29() to its fixed point (x < 13.2 p, y, zz, zzz < 2 p, every limb < 2^32, every
// column sum < 2^64) and cross-checks the limb algorithms against big integers.
#pragma once
#include "curve.hpp"
#include "ff29_mul_gfx950.inc"

namespace zkmi {

struct U29 {
    uint32_t l[9];
};

struct Fp29 {
    static constexpr uint32_t MASK = 0x1fffffffu;
    static constexpr uint32_t P[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    static constexpr uint32_t NINV = 0x04866389u;  // -p^-1 mod 2^29
    // 2^261 mod p: "one" of the radix-2^261 Montgomery domain (normalised limbs)
    static constexpr uint32_t ONE[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
    // k*p with every limb >= 2^30 (the top limb takes the rest): a + BIAS_k - b has no negative limb for a weakly normalised b < k*p
    static constexpr uint32_t BIAS4[9] = {0x41f3f51cu, 0x441182d9u, 0x51ca8d3au, 0x4b548b41u, 0x561765deu, 0x4b6d0300u, 0x429b8502u, 0x597098ceu, 0x00c19137u};
    static constexpr uint32_t BIAS8[9] = {0x43e7ea38u, 0x482305b4u, 0x43951a76u, 0x56a91685u, 0x4c2ecbbeu, 0x56da0603u, 0x45370a06u, 0x52e1319eu, 0x01832271u};
    static constexpr uint32_t BIAS12[9] = {0x45dbdf54u, 0x4c34888fu, 0x555fa7b2u, 0x41fda1c8u, 0x4246319fu, 0x42470906u, 0x47d28f0bu, 0x4c51ca6eu, 0x0244b3abu};
    static constexpr uint32_t BIAS16[9] = {0x47cfd470u, 0x50460b6au, 0x472a34eeu, 0x4d522d0cu, 0x585d977fu, 0x4db40c08u, 0x4a6e140fu, 0x45c2633eu, 0x030644e5u};
    static constexpr uint32_t BIAS24[9] = {0x4bb7bea8u, 0x58691120u, 0x4abf4f66u, 0x43fb4393u, 0x448c6340u, 0x448e120eu, 0x4fa51e18u, 0x58a394deu, 0x04896758u};
    static constexpr uint32_t BIAS40[9] = {0x53879318u, 0x48af1c8cu, 0x51e98457u, 0x514d70a1u, 0x5ce9fac1u, 0x52421e18u, 0x5a133229u, 0x5e65f81eu, 0x078fac3fu};
    static constexpr uint32_t BIAS32[9] = {0x4f9fa8e0u, 0x408c16d6u, 0x4e5469dfu, 0x5aa45a1au, 0x50bb2f00u, 0x5b681813u, 0x54dc2820u, 0x4b84c67eu, 0x060c89ccu};
    static constexpr uint32_t BIAS64[9] = {0x5f3f51c0u, 0x41182daeu, 0x5ca8d3c0u, 0x5548b436u, 0x41765e03u, 0x56d03029u, 0x49b85043u, 0x57098cffu, 0x0c19139au};
    static constexpr uint32_t BIAS80[9] = {0x470f2630u, 0x515e391bu, 0x43d308b0u, 0x429ae145u, 0x59d3f585u, 0x44843c33u, 0x54266455u, 0x5ccbf03fu, 0x0f1f5881u};
    static constexpr uint32_t PINV = 0x1b799c77u;  // p^-1 mod 2^29 (= 2^29 - NINV)
    static constexpr uint32_t RC[9] = {0x078302b9u, 0x1efb9f49u, 0x038d5cb0u, 0x1d2add2fu, 0x0a7a2687u, 0x1d24bf3fu, 0x1f591ebeu, 0x11a3d9cbu, 0x1fcf9bb1u};  // 2^261 - p
    static constexpr uint32_t QM = 0xa948e6d7u;    // floor(2^53 / ((p >> 232) + 1))
    template <int K>
    static constexpr uint32_t bias(int i) {
        static_assert(K == 4 || K == 8 || K == 12 || K == 16 || K == 24 || K == 32 || K == 40 || K == 64 || K == 80, "no bias table for this multiple of p");
        return K == 4 ? BIAS4[i] : K == 8 ? BIAS8[i] : K == 12 ? BIAS12[i] : K == 16 ? BIAS16[i] : K == 24 ? BIAS24[i] : K == 32 ? BIAS32[i]
             : K == 40 ? BIAS40[i] : K == 64 ? BIAS64[i] : BIAS80[i];
    }
};

__device__ __forceinline__ U29 u29_one() {
    U29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = Fp29::ONE[i];
    return r;
}

// a * b / 2^261 mod p (lazily reduced: < a*b/2^261 + p), limbs 0..7 < 2^29
__device__ __forceinline__ U29 u29_mul(const U29& a, const U29& b) {
    U29 r;
    asm(ZKMI_MONT_MUL29_ASM
        : [r0] "=&v"(r.l[0]), [r1] "=&v"(r.l[1]), [r2] "=&v"(r.l[2]), [r3] "=&v"(r.l[3]), [r4] "=&v"(r.l[4]), [r5] "=&v"(r.l[5]),
          [r6] "=&v"(r.l[6]), [r7] "=&v"(r.l[7]), [r8] "=&v"(r.l[8])
        : [a0] "v"(a.l[0]), [a1] "v"(a.l[1]), [a2] "v"(a.l[2]), [a3] "v"(a.l[3]), [a4] "v"(a.l[4]), [a5] "v"(a.l[5]), [a6] "v"(a.l[6]),
          [a7] "v"(a.l[7]), [a8] "v"(a.l[8]), [b0] "v"(b.l[0]), [b1] "v"(b.l[1]), [b2] "v"(b.l[2]), [b3] "v"(b.l[3]), [b4] "v"(b.l[4]),
          [b5] "v"(b.l[5]), [b6] "v"(b.l[6]), [b7] "v"(b.l[7]), [b8] "v"(b.l[8]), [p0] "s"(Fp29::P[0]), [p1] "s"(Fp29::P[1]),
          [p2] "s"(Fp29::P[2]), [p3] "s"(Fp29::P[3]), [p4] "s"(Fp29::P[4]), [p5] "s"(Fp29::P[5]), [p6] "s"(Fp29::P[6]), [p7] "s"(Fp29::P[7]),
          [p8] "s"(Fp29::P[8]), [ninv] "s"(Fp29::NINV)
        : "v0", "v1", "vcc");
    return r;
}

// a * a / 2^261 mod p: cross products once against the doubled operand (126 multiply-adds instead of 162); limbs of a < 2^31
__device__ __forceinline__ U29 u29_sqr(const U29& a) {
    U29 r, d;
#pragma unroll
    for (int i = 0; i < 9; i++) d.l[i] = a.l[i] + a.l[i];
    asm(ZKMI_MONT_SQR29_ASM
        : [r0] "=&v"(r.l[0]), [r1] "=&v"(r.l[1]), [r2] "=&v"(r.l[2]), [r3] "=&v"(r.l[3]), [r4] "=&v"(r.l[4]), [r5] "=&v"(r.l[5]),
          [r6] "=&v"(r.l[6]), [r7] "=&v"(r.l[7]), [r8] "=&v"(r.l[8])
        : [a0] "v"(a.l[0]), [a1] "v"(a.l[1]), [a2] "v"(a.l[2]), [a3] "v"(a.l[3]), [a4] "v"(a.l[4]), [a5] "v"(a.l[5]), [a6] "v"(a.l[6]),
          [a7] "v"(a.l[7]), [a8] "v"(a.l[8]), [d0] "v"(d.l[0]), [d1] "v"(d.l[1]), [d2] "v"(d.l[2]), [d3] "v"(d.l[3]), [d4] "v"(d.l[4]),
          [d5] "v"(d.l[5]), [d6] "v"(d.l[6]), [d7] "v"(d.l[7]), [d8] "v"(d.l[8]), [p0] "s"(Fp29::P[0]), [p1] "s"(Fp29::P[1]),
          [p2] "s"(Fp29::P[2]), [p3] "s"(Fp29::P[3]), [p4] "s"(Fp29::P[4]), [p5] "s"(Fp29::P[5]), [p6] "s"(Fp29::P[6]), [p7] "s"(Fp29::P[7]),
          [p8] "s"(Fp29::P[8]), [ninv] "s"(Fp29::NINV)
        : "v0", "v1", "vcc");
    return r;
}

__device__ __forceinline__ U29 u29_wnorm_fwd(const U29& a) {
    U29 r;
    r.l[0] = a.l[0] & 0x1fffffffu;
#pragma unroll
    for (int i = 1; i < 8; i++) r.l[i] = (a.l[i] & 0x1fffffffu) + (a.l[i - 1] >> 29);
    r.l[8] = a.l[8] + (a.l[7] >> 29);
    return r;
}

// (a0*b0 + a1*b1 [+ a2*b2 [+ a3*b3]]) / 2^261 mod p with ONE Montgomery reduction: the column accumulators take all products.
#define ZK29_IN(t, A, B)                                                                                                              \
    [a##t##_0] "v"(A.l[0]), [a##t##_1] "v"(A.l[1]), [a##t##_2] "v"(A.l[2]), [a##t##_3] "v"(A.l[3]), [a##t##_4] "v"(A.l[4]),          \
        [a##t##_5] "v"(A.l[5]), [a##t##_6] "v"(A.l[6]), [a##t##_7] "v"(A.l[7]), [a##t##_8] "v"(A.l[8]), [b##t##_0] "v"(B.l[0]),      \
        [b##t##_1] "v"(B.l[1]), [b##t##_2] "v"(B.l[2]), [b##t##_3] "v"(B.l[3]), [b##t##_4] "v"(B.l[4]), [b##t##_5] "v"(B.l[5]),      \
        [b##t##_6] "v"(B.l[6]), [b##t##_7] "v"(B.l[7]), [b##t##_8] "v"(B.l[8])
#define ZK29_OUT(r)                                                                                                                   \
    [r0] "=&v"(r.l[0]), [r1] "=&v"(r.l[1]), [r2] "=&v"(r.l[2]), [r3] "=&v"(r.l[3]), [r4] "=&v"(r.l[4]), [r5] "=&v"(r.l[5]),          \
        [r6] "=&v"(r.l[6]), [r7] "=&v"(r.l[7]), [r8] "=&v"(r.l[8])
#define ZK29_SG                                                                                                                       \
    [p0] "s"(Fp29::P[0]), [p1] "s"(Fp29::P[1]), [p2] "s"(Fp29::P[2]), [p3] "s"(Fp29::P[3]), [p4] "s"(Fp29::P[4]), [p5] "s"(Fp29::P[5]), \
        [p6] "s"(Fp29::P[6]), [p7] "s"(Fp29::P[7]), [p8] "s"(Fp29::P[8]), [ninv] "s"(Fp29::NINV)

__device__ __forceinline__ U29 u29_mul2(const U29& a0, const U29& b0, const U29& a1, const U29& b1) {
    U29 r;
    asm(ZKMI_MONT_MUL29_N2_ASM : ZK29_OUT(r) : ZK29_IN(0, a0, b0), ZK29_IN(1, a1, b1), ZK29_SG : "v0", "v1", "vcc");
    return r;
}
__device__ __forceinline__ U29 u29_mul3(const U29& a0, const U29& b0, const U29& a1, const U29& b1, const U29& a2, const U29& b2) {
    U29 r;
    asm(ZKMI_MONT_MUL29_N3_ASM : ZK29_OUT(r) : ZK29_IN(0, a0, b0), ZK29_IN(1, a1, b1), ZK29_IN(2, a2, b2), ZK29_SG : "v0", "v1", "vcc");
    return r;
}
__device__ __forceinline__ U29 u29_mul4(const U29& a0, const U29& b0, const U29& a1, const U29& b1, const U29& a2, const U29& b2, const U29& a3,
                                        const U29& b3) {
    U29 r;
    asm(ZKMI_MONT_MUL29_N4_ASM
        : ZK29_OUT(r)
        : ZK29_IN(0, a0, b0), ZK29_IN(1, a1, b1), ZK29_IN(2, a2, b2), ZK29_IN(3, a3, b3), ZK29_SG
        : "v0", "v1", "vcc");
    return r;
}

// Two INDEPENDENT products (squarings) in one block, instruction streams interleaved (gen_ff29_asm.py emit_pair): same instructions as two u29_mul,
// but the wave has two multiply-add chains in flight instead of one.
#define ZK29_X2_IN(t, N, A)                                                                                                                \
    [N##0_##t] "v"(A.l[0]), [N##1_##t] "v"(A.l[1]), [N##2_##t] "v"(A.l[2]), [N##3_##t] "v"(A.l[3]), [N##4_##t] "v"(A.l[4]),                \
        [N##5_##t] "v"(A.l[5]), [N##6_##t] "v"(A.l[6]), [N##7_##t] "v"(A.l[7]), [N##8_##t] "v"(A.l[8])
#define ZK29_X2_OUT(t, r)                                                                                                                  \
    [r0_##t] "=&v"(r.l[0]), [r1_##t] "=&v"(r.l[1]), [r2_##t] "=&v"(r.l[2]), [r3_##t] "=&v"(r.l[3]), [r4_##t] "=&v"(r.l[4]),                 \
        [r5_##t] "=&v"(r.l[5]), [r6_##t] "=&v"(r.l[6]), [r7_##t] "=&v"(r.l[7]), [r8_##t] "=&v"(r.l[8])
__device__ __forceinline__ void u29_mul_x2(const U29& a0, const U29& b0, const U29& a1, const U29& b1, U29& r0, U29& r1) {
    asm(ZKMI_MONT_MUL29_X2_ASM
        : ZK29_X2_OUT(0, r0), ZK29_X2_OUT(1, r1)
        : ZK29_X2_IN(0, a, a0), ZK29_X2_IN(0, b, b0), ZK29_X2_IN(1, a, a1), ZK29_X2_IN(1, b, b1), ZK29_SG
        : "v0", "v1", "v2", "v3", "vcc");
}
__device__ __forceinline__ void u29_sqr_x2(const U29& a0, const U29& a1, U29& r0, U29& r1) {
    U29 d0, d1;
#pragma unroll
    for (int i = 0; i < 9; i++) { d0.l[i] = a0.l[i] + a0.l[i]; d1.l[i] = a1.l[i] + a1.l[i]; }
    asm(ZKMI_MONT_SQR29_X2_ASM
        : ZK29_X2_OUT(0, r0), ZK29_X2_OUT(1, r1)
        : ZK29_X2_IN(0, a, a0), ZK29_X2_IN(0, d, d0), ZK29_X2_IN(1, a, a1), ZK29_X2_IN(1, d, d1), ZK29_SG
        : "v0", "v1", "v2", "v3", "vcc");
}

// K*p - a, weakly normalised  (a weakly normalised, < K*p)
template <int K>
__device__ __forceinline__ U29 u29_neg(const U29& a) {
    U29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = Fp29::bias<K>(i) - a.l[i];
    return u29_wnorm_fwd(r);
}

__device__ __forceinline__ U29 u29_add(const U29& a, const U29& b) {
    U29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}

// a - b + K*p  (b must be weakly normalised and < K*p)
template <int K>
__device__ __forceinline__ U29 u29_sub(const U29& a, const U29& b) {
    U29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        r.l[i] = a.l[i] - b.l[i] + Fp29::bias<K>(i);
    }
    return r;
}

// one parallel carry pass: limbs 0..7 back below 2^29 + 8
__device__ __forceinline__ U29 u29_wnorm(const U29& a) {
    U29 r;
    r.l[0] = a.l[0] & Fp29::MASK;
#pragma unroll
    for (int i = 1; i < 8; i++) r.l[i] = (a.l[i] & Fp29::MASK) + (a.l[i - 1] >> 29);
    r.l[8] = a.l[8] + (a.l[7] >> 29);
    return r;
}

// exact carry propagation (serial): limbs 0..7 < 2^29
__device__ __forceinline__ U29 u29_ripple(const U29& a) {
    U29 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t t = a.l[i] + c;
        r.l[i] = t & Fp29::MASK;
        c = t >> 29;
    }
    r.l[8] = a.l[8] + c;
    return r;
}

// canonical Montgomery-2^256 image (8 x u32, < p) -> V << 5 in 29-bit limbs
__device__ __forceinline__ U29 u29_load(const Fp& v) {
    U29 r;
    const uint32_t* w = v.l;
    r.l[0] = (w[0] << 5) & Fp29::MASK;
    r.l[1] = __funnelshift_r(w[0], w[1], 24) & Fp29::MASK;
    r.l[2] = __funnelshift_r(w[1], w[2], 21) & Fp29::MASK;
    r.l[3] = __funnelshift_r(w[2], w[3], 18) & Fp29::MASK;
    r.l[4] = __funnelshift_r(w[3], w[4], 15) & Fp29::MASK;
    r.l[5] = __funnelshift_r(w[4], w[5], 12) & Fp29::MASK;
    r.l[6] = __funnelshift_r(w[5], w[6], 9) & Fp29::MASK;
    r.l[7] = __funnelshift_r(w[6], w[7], 6) & Fp29::MASK;
    r.l[8] = w[7] >> 3;
    return r;
}

// 8 x u32 image (canonical, or a lazily reduced intermediate < 2^256) -> limbs, no shift
template <class E>
__device__ __forceinline__ U29 u29_unpack(const E& v) {
    U29 r;
    const uint32_t* w = v.l;
    r.l[0] = w[0] & 0x1fffffffu;
    r.l[1] = __funnelshift_r(w[0], w[1], 29) & 0x1fffffffu;
    r.l[2] = __funnelshift_r(w[1], w[2], 26) & 0x1fffffffu;
    r.l[3] = __funnelshift_r(w[2], w[3], 23) & 0x1fffffffu;
    r.l[4] = __funnelshift_r(w[3], w[4], 20) & 0x1fffffffu;
    r.l[5] = __funnelshift_r(w[4], w[5], 17) & 0x1fffffffu;
    r.l[6] = __funnelshift_r(w[5], w[6], 14) & 0x1fffffffu;
    r.l[7] = __funnelshift_r(w[6], w[7], 11) & 0x1fffffffu;
    r.l[8] = w[7] >> 8;
    return r;
}

// any lazily reduced value (< 2^260) -> canonical Montgomery-2^256 image
__device__ __forceinline__ Fp u29_store(const U29& x) {
    U29 c;
#pragma unroll
    for (int i = 0; i < 8; i++) c.l[i] = 0;
    c.l[8] = 1u << 24;  // 2^256
    U29 t = u29_ripple(u29_mul(x, c));  // = x * 2^256 / 2^261 ; < x/32 + p < 10 p for every x this file produces
    uint32_t w[8];
    w[0] = t.l[0] | (t.l[1] << 29);
    w[1] = (t.l[1] >> 3) | (t.l[2] << 26);
    w[2] = (t.l[2] >> 6) | (t.l[3] << 23);
    w[3] = (t.l[3] >> 9) | (t.l[4] << 20);
    w[4] = (t.l[4] >> 12) | (t.l[5] << 17);
    w[5] = (t.l[5] >> 15) | (t.l[6] << 14);
    w[6] = (t.l[6] >> 18) | (t.l[7] << 11);
    w[7] = (t.l[7] >> 21) | (t.l[8] << 8);
    // every value this file stores is < 13.5 p, so t < 1.5 p; handle t < 4 p anyway: conditional subtractions of 2p, then p
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = w[i];
#pragma unroll
    for (int k = 1; k >= 0; k--) {
        uint32_t s[8];
        uint64_t bw = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint32_t pk = (k == 0) ? FpParams::MOD[i] : ((FpParams::MOD[i] << 1) | (i ? (FpParams::MOD[i - 1] >> 31) : 0u));
            uint64_t d = (uint64_t)r.l[i] - pk - bw;
            s[i] = (uint32_t)d;
            bw = d >> 63;
        }
        if (!bw) {
#pragma unroll
            for (int i = 0; i < 8; i++) r.l[i] = s[i];
        }
    }
    return r;
}

// x is the direct output of u29_mul with a product bound < p*2^261 (so x < 2p): x == 0 mod p ?
__device__ __forceinline__ bool u29_mulout_is_zero(const U29& x) {
    U29 t = u29_ripple(x);
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        z |= t.l[i];
        e |= t.l[i] ^ Fp29::P[i];
    }
    return z == 0 || e == 0;
}

// XYZZ accumulator in the unsaturated representation (G1)
struct Acc29 {
    U29 x, y, zz, zzz;
    bool inf;
};

__device__ __forceinline__ void acc29_from_xyzz(Acc29& A, const XYZZ<Fp>& c) {
    A.inf = c.is_inf();
    const U29 one = u29_one();
    A.x = u29_mul(u29_load(c.x), one);  // contracting multiplication: value < 1.2 p
    A.y = u29_mul(u29_load(c.y), one);
    A.zz = u29_mul(u29_load(c.zz), one);
    A.zzz = u29_mul(u29_load(c.zzz), one);
}
__device__ __forceinline__ XYZZ<Fp> acc29_to_xyzz(const Acc29& A) {
    if (A.inf) return XYZZ<Fp>::inf();
    return XYZZ<Fp>{u29_store(A.x), u29_store(A.y), u29_store(A.zz), u29_store(A.zzz)};
}

// A += (px, py), an affine point in gnark's memory image (sign already applied).  madd-2008-s; the operation order and
// bias multiples are exactly those of tools/u29_model.py::madd_fp.
__device__ __forceinline__ void xyzz_madd29(Acc29& A, const Fp& px, const Fp& py) {
    if (px.is_zero() && py.is_zero()) return;
    const U29 x2 = u29_load(px), y2 = u29_load(py);
    if (A.inf) {
        const U29 one = u29_one();
        A.x = u29_mul(x2, one);
        A.y = u29_mul(y2, one);
        A.zz = one;
        A.zzz = one;
        A.inf = false;
        return;
    }
#ifdef ZKMI_MUL_PAIR
    // the same ten products, eight of them issued as four interleaved pairs of INDEPENDENT ones: (U2, S2), (P^2, R^2), (ZZ1 PP, P PP), (X1 PP, ZZZ1 PPP)
    U29 U2, S2;
    u29_mul_x2(x2, A.zz, y2, A.zzz, U2, S2);
    const U29 P = u29_wnorm(u29_sub<16>(U2, A.x));
    const U29 R = u29_wnorm(u29_sub<4>(S2, A.y));
    U29 PP, RR;
    u29_sqr_x2(P, R, PP, RR);
    U29 ZZ3, PPP;
    u29_mul_x2(A.zz, PP, P, PP, ZZ3, PPP);
    if ((((ZZ3.l[0] & Fp29::MASK) * Fp29::PINV) & Fp29::MASK) < 2u) {
        XYZZ<Fp> c = acc29_to_xyzz(A);
        c.madd(px, py);
        acc29_from_xyzz(A, c);
        return;
    }
    {
        U29 Q, ZZZ3;
        u29_mul_x2(A.x, PP, A.zzz, PPP, Q, ZZZ3);
        U29 t = u29_wnorm(u29_sub<4>(RR, PPP));
        t = u29_sub<4>(t, Q);
        t = u29_sub<4>(t, Q);
        const U29 X3 = u29_wnorm(t);
        const U29 d = u29_wnorm(u29_sub<16>(Q, X3));
        const U29 Y3 = u29_mul2(R, d, u29_neg<4>(A.y), PPP);
        A.zzz = ZZZ3;
        A.zz = ZZ3;
        A.x = X3;
        A.y = Y3;
        return;
    }
#else
    const U29 U2 = u29_mul(x2, A.zz);
    const U29 S2 = u29_mul(y2, A.zzz);
    const U29 P = u29_wnorm(u29_sub<16>(U2, A.x));
    const U29 R = u29_wnorm(u29_sub<4>(S2, A.y));
    const U29 PP = u29_sqr(P);
    const U29 ZZ3 = u29_mul(A.zz, PP);
    if ((((ZZ3.l[0] & Fp29::MASK) * Fp29::PINV) & Fp29::MASK) < 2u) {
        // ZZ3 = ZZ1 * P^2 is a product output < 2 p: it can only be == 0 mod p (P == 0: same x, i.e. doubling or P + (-P)) if
        // ZZ3 / p mod 2^29 is 0 or 1 -- a three-instruction filter that never misses and fires wrongly once in 2^28.  Rare either
        // way: take the canonical saturated path, which is correct for every input (inline: an out-of-line call here costs the G1
        // kernel 25 VGPRs and a wave of occupancy).
        XYZZ<Fp> c = acc29_to_xyzz(A);
        c.madd(px, py);
        acc29_from_xyzz(A, c);
        return;
    }
    const U29 PPP = u29_mul(P, PP);
    const U29 Q = u29_mul(A.x, PP);
    U29 t = u29_sqr(R);
    t = u29_wnorm(u29_sub<4>(t, PPP));
    t = u29_sub<4>(t, Q);
    t = u29_sub<4>(t, Q);
    const U29 X3 = u29_wnorm(t);
    const U29 d = u29_wnorm(u29_sub<16>(Q, X3));
    const U29 Y3 = u29_mul2(R, d, u29_neg<4>(A.y), PPP);  // R d - Y1 PPP under ONE reduction: a direct product output (< 2 p)
    A.zzz = u29_mul(A.zzz, PPP);
    A.zz = ZZ3;
    A.x = X3;
    A.y = Y3;
#endif
}

// m < 3p is a direct product output: m == 0 mod p ?
__device__ __forceinline__ bool u29_mulout3_is_zero(const U29& x) {
    U29 t = u29_ripple(x);
    uint32_t z = 0, e1 = 0, e2 = 0;
    // 2p in normalised limbs
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        uint32_t p2 = 2 * Fp29::P[i] + c;
        c = (i < 8) ? (p2 >> 29) : 0;
        if (i < 8) p2 &= Fp29::MASK;
        z |= t.l[i];
        e1 |= t.l[i] ^ Fp29::P[i];
        e2 |= t.l[i] ^ p2;
    }
    return z == 0 || e1 == 0 || e2 == 0;
}

// ---- interchange format between the G1 kernels (task partials, reduction levels): the lazily reduced value itself, made canonical
// and packed into 8 words -- i.e. x * 2^261 mod p, NOT gnark's x * 2^256 image.  Writing it needs a partial reduction (x -= q p, q from
// the top limb) instead of a Montgomery multiplication, reading it is a plain unpack (value < p); the host divides the handful of
// final sums by 2^5 (msm_finish).  Infinity stays zz == 0.
__device__ __forceinline__ U29 u29p_reduce(const U29& x) {  // x weakly normalised -> < 2.01 p, limbs 0..7 < 2^29 (as u29r_reduce, modulus p)
    const uint32_t q = __umulhi(x.l[8], Fp29::QM) >> 21;
    U29 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t acc = (uint64_t)q * Fp29::RC[i] + (uint64_t)(x.l[i] + c);
        r.l[i] = (uint32_t)acc & Fp29::MASK;
        c = (uint32_t)(acc >> 29);
    }
    uint64_t acc = (uint64_t)q * Fp29::RC[8] + (uint64_t)(x.l[8] + c);
    r.l[8] = (uint32_t)(acc - ((uint64_t)q << 29));
    return r;
}
__device__ __forceinline__ Fp u29p_pack(const U29& x) {  // any weakly normalised value (top limb < 2^32) -> canonical, packed
    const U29 t = u29p_reduce(x);
    Fp r;
    r.l[0] = t.l[0] | (t.l[1] << 29);
    r.l[1] = (t.l[1] >> 3) | (t.l[2] << 26);
    r.l[2] = (t.l[2] >> 6) | (t.l[3] << 23);
    r.l[3] = (t.l[3] >> 9) | (t.l[4] << 20);
    r.l[4] = (t.l[4] >> 12) | (t.l[5] << 17);
    r.l[5] = (t.l[5] >> 15) | (t.l[6] << 14);
    r.l[6] = (t.l[6] >> 18) | (t.l[7] << 11);
    r.l[7] = (t.l[7] >> 21) | (t.l[8] << 8);
#pragma unroll
    for (int k = 0; k < 2; k++) {
        uint32_t s[8];
        uint64_t bw = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t d = (uint64_t)r.l[i] - FpParams::MOD[i] - bw;
            s[i] = (uint32_t)d;
            bw = d >> 63;
        }
        if (!bw) {
#pragma unroll
            for (int i = 0; i < 8; i++) r.l[i] = s[i];
        }
    }
    return r;
}
__device__ __forceinline__ XYZZ<Fp> acc29_to_packed(const Acc29& A) {
    if (A.inf) return XYZZ<Fp>::inf();
    return XYZZ<Fp>{u29p_pack(A.x), u29p_pack(A.y), u29p_pack(A.zz), u29p_pack(A.zzz)};
}

// ---- XYZZ + XYZZ and doubling for the bucket-reduction tail (G1).  Class invariant proven by tools/u29_model.py
// (check_add_dbl_class): with every input coordinate < 32 p and weakly normalised, every output coordinate is again < 32 p.
__device__ __forceinline__ void acc29_load(Acc29& A, const XYZZ<Fp>& c) {  // interchange format (acc29_to_packed) -> limbs, value < p
    A.inf = c.is_inf();
    A.x = u29_unpack(c.x);
    A.y = u29_unpack(c.y);
    A.zz = u29_unpack(c.zz);
    A.zzz = u29_unpack(c.zzz);
}
__device__ __forceinline__ void acc29_dbl(Acc29& A) {  // dbl-2008-s-1
    if (A.inf) return;
    const U29 U = u29_add(A.y, A.y);
    const U29 V = u29_sqr(U);
    const U29 Wv = u29_mul(U, V);
    const U29 S = u29_mul(A.x, V);
    const U29 X2 = u29_sqr(A.x);
    const U29 M = u29_wnorm(u29_add(u29_add(X2, X2), X2));
    U29 t = u29_sqr(M);
    t = u29_sub<8>(t, S);
    const U29 X3 = u29_wnorm(u29_sub<8>(t, S));
    const U29 d = u29_wnorm(u29_sub<24>(S, X3));
    const U29 Y3 = u29_mul2(M, d, u29_neg<12>(Wv), A.y);
    A.zz = u29_mul(V, A.zz);
    A.zzz = u29_mul(Wv, A.zzz);
    A.x = X3;
    A.y = Y3;
}
__device__ __forceinline__ void acc29_add(Acc29& A, const Acc29& Bq) {  // add-2008-s
    if (Bq.inf) return;
    if (A.inf) { A = Bq; return; }
    const U29 U1 = u29_mul(A.x, Bq.zz), U2 = u29_mul(Bq.x, A.zz);
    const U29 S1 = u29_mul(A.y, Bq.zzz), S2 = u29_mul(Bq.y, A.zzz);
    const U29 P = u29_wnorm(u29_sub<8>(U2, U1));
    const U29 R = u29_wnorm(u29_sub<8>(S2, S1));
    const U29 PP = u29_sqr(P);
    if (u29_mulout3_is_zero(PP)) {
        // same x: doubling or P + (-P).  R = S2 - S1 decides; R*R is a product output (< 3 p), so the same test applies.
        if (u29_mulout3_is_zero(u29_sqr(R))) acc29_dbl(A);
        else A.inf = true;
        return;
    }
    const U29 PPP = u29_mul(P, PP);
    const U29 Q = u29_mul(U1, PP);
    U29 t = u29_sqr(R);
    t = u29_wnorm(u29_sub<4>(t, PPP));
    t = u29_sub<4>(t, Q);
    t = u29_sub<4>(t, Q);
    const U29 X3 = u29_wnorm(t);
    const U29 d = u29_wnorm(u29_sub<16>(Q, X3));
    const U29 Y3 = u29_mul2(R, d, u29_neg<8>(S1), PPP);
    A.zz = u29_mul(u29_mul(A.zz, Bq.zz), PP);
    A.zzz = u29_mul(u29_mul(A.zzz, Bq.zzz), PPP);
    A.x = X3;
    A.y = Y3;
}

// ---- the same two operations spread over a QUAD of lanes: a point addition is 14 products of dependency depth 4 (a doubling: 10 of
// depth 3), and the last reduction tail of a proof runs on an otherwise idle machine where only the LENGTH of the serial chain counts.
// All four lanes of a quad hold the same operands; in every round each lane multiplies a different pair and the four products are
// broadcast back by DPP quad permutes.  Same formulas, same bias multiples, same bounds as acc29_add / acc29_dbl above.
template <int K>
__device__ __forceinline__ U29 u29_quad_bcast(const U29& x) {  // the value held by lane K of the quad, in all four lanes
    U29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)x.l[i], K | (K << 2) | (K << 4) | (K << 6), 0xf, 0xf, false);
    return r;
}
__device__ __forceinline__ U29 u29_sel4(unsigned q, const U29& a0, const U29& a1, const U29& a2, const U29& a3) {
    // by masks, not by ?: -- hipcc turns a chain of conditional member reads into a select of ADDRESSES and parks the operands in scratch
    const uint32_t m0 = 0u - (uint32_t)(q == 0), m1 = 0u - (uint32_t)(q == 1), m2 = 0u - (uint32_t)(q == 2), m3 = 0u - (uint32_t)(q == 3);
    U29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = (a0.l[i] & m0) | (a1.l[i] & m1) | (a2.l[i] & m2) | (a3.l[i] & m3);
    return r;
}
__device__ __forceinline__ void acc29_dbl_quad(Acc29& A, unsigned q) {
    if (A.inf) return;
    const U29 U = u29_add(A.y, A.y);
    U29 m = u29_mul(u29_sel4(q, U, A.x, U, U), u29_sel4(q, U, A.x, U, U));                    // round 1: U*U, X*X
    const U29 V = u29_quad_bcast<0>(m), X2 = u29_quad_bcast<1>(m);
    const U29 M = u29_wnorm(u29_add(u29_add(X2, X2), X2));
    m = u29_mul(u29_sel4(q, U, A.x, M, V), u29_sel4(q, V, V, M, A.zz));                          // round 2: U*V, X*V, M*M, V*ZZ
    const U29 Wv = u29_quad_bcast<0>(m), S = u29_quad_bcast<1>(m), ZZ3 = u29_quad_bcast<3>(m);
    U29 t = u29_quad_bcast<2>(m);
    t = u29_sub<8>(t, S);
    const U29 X3 = u29_wnorm(u29_sub<8>(t, S));
    const U29 d = u29_wnorm(u29_sub<24>(S, X3));
    m = u29_mul(u29_sel4(q, M, Wv, Wv, M), u29_sel4(q, d, A.y, A.zzz, d));                        // round 3: M*d, W*Y, W*ZZZ
    A.y = u29_wnorm(u29_sub<4>(u29_quad_bcast<0>(m), u29_quad_bcast<1>(m)));
    A.zzz = u29_quad_bcast<2>(m);
    A.zz = ZZ3;
    A.x = X3;
}
__device__ __forceinline__ void acc29_add_quad(Acc29& A, const Acc29& Bq, unsigned q) {
    if (Bq.inf) return;
    if (A.inf) { A = Bq; return; }
    U29 m = u29_mul(u29_sel4(q, A.x, Bq.x, A.y, Bq.y), u29_sel4(q, Bq.zz, A.zz, Bq.zzz, A.zzz));  // round 1: U1, U2, S1, S2
    const U29 U1 = u29_quad_bcast<0>(m), S1 = u29_quad_bcast<2>(m);
    const U29 P = u29_wnorm(u29_sub<8>(u29_quad_bcast<1>(m), U1));
    const U29 R = u29_wnorm(u29_sub<8>(u29_quad_bcast<3>(m), S1));
    m = u29_mul(u29_sel4(q, P, R, A.zz, A.zzz), u29_sel4(q, P, R, Bq.zz, Bq.zzz));                // round 2: P*P, R*R, ZZ1*ZZ2, ZZZ1*ZZZ2
    const U29 PP = u29_quad_bcast<0>(m), RR = u29_quad_bcast<1>(m), ZZ12 = u29_quad_bcast<2>(m), ZZZ12 = u29_quad_bcast<3>(m);
    if (u29_mulout3_is_zero(PP)) {
        // same x: doubling or P + (-P); R*R is a product output (< 3 p), so the same test applies (uniform over the quad)
        if (u29_mulout3_is_zero(RR)) acc29_dbl_quad(A, q);
        else A.inf = true;
        return;
    }
    m = u29_mul(u29_sel4(q, P, U1, ZZ12, P), PP);                                                  // round 3: P*PP, U1*PP, ZZ12*PP
    const U29 PPP = u29_quad_bcast<0>(m), Q = u29_quad_bcast<1>(m), ZZ3 = u29_quad_bcast<2>(m);
    U29 t = u29_wnorm(u29_sub<4>(RR, PPP));
    t = u29_sub<4>(t, Q);
    t = u29_sub<4>(t, Q);
    const U29 X3 = u29_wnorm(t);
    const U29 d = u29_wnorm(u29_sub<16>(Q, X3));
    m = u29_mul(u29_sel4(q, R, S1, ZZZ12, R), u29_sel4(q, d, PPP, PPP, d));                        // round 4: R*d, S1*PPP, ZZZ12*PPP
    A.y = u29_wnorm(u29_sub<4>(u29_quad_bcast<0>(m), u29_quad_bcast<1>(m)));
    A.zzz = u29_quad_bcast<2>(m);
    A.zz = ZZ3;
    A.x = X3;
}

// ------------------------------------------------------------------------------------------------------------ G2 (Fp2)
// Components are lazily reduced U29 values.  Operation order, bias multiples and the two contractions are exactly those of
// tools/u29_model.py::madd_fp2, whose bound propagation closes at x, y < 2 p, zz, zzz < 10.6 p.
struct U29x2 {
    U29 c0, c1;
};

__device__ __forceinline__ U29x2 f2_mul29(const U29x2& a, const U29x2& b) {
    const U29 v0 = u29_mul(a.c0, b.c0), v1 = u29_mul(a.c1, b.c1);
    const U29 s = u29_mul(u29_add(a.c0, a.c1), u29_add(b.c0, b.c1));
    U29x2 r;
    r.c0 = u29_wnorm(u29_sub<4>(v0, v1));
    r.c1 = u29_wnorm(u29_sub<4>(u29_sub<4>(s, v0), v1));
    return r;
}
// (a0+a1)(a0-a1), 2 a0 a1 ; *m_out = a0*a1 (a direct product output: cheap zero filter for the caller)
template <int KD>
__device__ __forceinline__ U29x2 f2_sqr29(const U29x2& a, U29* m_out) {
    const U29 d = u29_wnorm(u29_sub<KD>(a.c0, a.c1));
    const U29 m = u29_mul(a.c0, a.c1);
    U29x2 r;
    r.c0 = u29_mul(u29_add(a.c0, a.c1), d);
    r.c1 = u29_add(m, m);
    if (m_out) *m_out = m;
    return r;
}
template <int K>
__device__ __forceinline__ U29x2 f2_sub29(const U29x2& a, const U29x2& b) {
    return U29x2{u29_wnorm(u29_sub<K>(a.c0, b.c0)), u29_wnorm(u29_sub<K>(a.c1, b.c1))};
}
__device__ __forceinline__ U29x2 f2_contract29(const U29x2& a) {
    const U29 one = u29_one();
    return U29x2{u29_mul(a.c0, one), u29_mul(a.c1, one)};
}
__device__ __forceinline__ U29x2 f2_load29(const Fp2& v) { return U29x2{u29_load(v.a0), u29_load(v.a1)}; }
__device__ __forceinline__ Fp2 f2_store29(const U29x2& v) { return Fp2{u29_store(v.c0), u29_store(v.c1)}; }

struct Acc29G2 {
    U29x2 x, y, zz, zzz;
    bool inf;
};
__device__ __forceinline__ void acc29g2_from_xyzz(Acc29G2& A, const XYZZ<Fp2>& c) {
    A.inf = c.is_inf();
    A.x = f2_contract29(f2_load29(c.x));
    A.y = f2_contract29(f2_load29(c.y));
    A.zz = f2_contract29(f2_load29(c.zz));
    A.zzz = f2_contract29(f2_load29(c.zzz));
}
__device__ __forceinline__ XYZZ<Fp2> acc29g2_to_xyzz(const Acc29G2& A) {
    if (A.inf) return XYZZ<Fp2>::inf();
    return XYZZ<Fp2>{f2_store29(A.x), f2_store29(A.y), f2_store29(A.zz), f2_store29(A.zzz)};
}

// Fp2 product with one reduction per component: c0 = a0 b0 + (-a1) b1, c1 = a0 b1 + a1 b0  (na1 = K*p - a1)
__device__ __forceinline__ U29x2 f2_mulF29(const U29x2& a, const U29x2& b, const U29& na1) {
    return U29x2{u29_mul2(a.c0, b.c0, na1, b.c1), u29_mul2(a.c0, b.c1, a.c1, b.c0)};
}

// A += (px, py): fused multi-product schedule of tools/u29_model.py::madd_fp2_fused -- every stored coordinate is a direct
// product output (< 2 p), no contraction multiplications:  X3 = R^2 - (P + 2 X1) PP,  Y3 = R (Q - X3) - Y1 PPP.
__device__ __forceinline__ void xyzz_madd29(Acc29G2& A, const Fp2& px, const Fp2& py) {
    if (px.is_zero() && py.is_zero()) return;
    const U29x2 x2 = f2_load29(px), y2 = f2_load29(py);
    if (A.inf) {
        A.x = f2_contract29(x2);
        A.y = f2_contract29(y2);
        U29 z;
#pragma unroll
        for (int i = 0; i < 9; i++) z.l[i] = 0;
        A.zz = U29x2{u29_one(), z};
        A.zzz = A.zz;
        A.inf = false;
        return;
    }
    const U29x2 U2 = f2_mulF29(x2, A.zz, u29_neg<40>(x2.c1));
    const U29x2 S2 = f2_mulF29(y2, A.zzz, u29_neg<40>(y2.c1));
    const U29x2 P = f2_sub29<4>(U2, A.x);
    const U29x2 R = f2_sub29<4>(S2, A.y);
    const U29 nP1 = u29_neg<8>(P.c1), nR1 = u29_neg<8>(R.c1);
    U29x2 PP;  // complex squaring: ((P0 + P1)(P0 - P1), 2 P0 P1) -- two products
    PP.c0 = u29_mul(u29_add(P.c0, P.c1), u29_wnorm(u29_sub<8>(P.c0, P.c1)));
    PP.c1 = u29_mul(u29_add(P.c0, P.c0), P.c1);
    if (u29_mulout3_is_zero(PP.c1)) {
        // 2 P0 P1 == 0 mod p is NECESSARY for P == 0 (same x: doubling or P + (-P)); the canonical saturated path is correct
        // for every input, so it is taken whenever the filter fires.
        XYZZ<Fp2> c = acc29g2_to_xyzz(A);
        c.madd(px, py);
        acc29g2_from_xyzz(A, c);
        return;
    }
    const U29x2 PPP = f2_mulF29(P, PP, nP1);
    const U29x2 Q = f2_mulF29(A.x, PP, u29_neg<8>(A.x.c1));
    const U29 W0 = u29_wnorm(u29_add(P.c0, u29_add(A.x.c0, A.x.c0))), W1 = u29_wnorm(u29_add(P.c1, u29_add(A.x.c1, A.x.c1)));
    const U29 nW0 = u29_neg<12>(W0), nW1 = u29_neg<12>(W1);
    U29x2 X3;
    X3.c0 = u29_mul3(u29_add(R.c0, R.c1), u29_wnorm(u29_sub<8>(R.c0, R.c1)), nW0, PP.c0, W1, PP.c1);  // R0^2 - R1^2 as ONE product
    X3.c1 = u29_mul3(u29_add(R.c0, R.c0), R.c1, nW0, PP.c1, nW1, PP.c0);
    const U29x2 d = f2_sub29<4>(Q, X3);
    const U29 nY0 = u29_neg<8>(A.y.c0), nY1 = u29_neg<8>(A.y.c1);
    U29x2 Y3;
    Y3.c0 = u29_mul4(R.c0, d.c0, nR1, d.c1, nY0, PPP.c0, A.y.c1, PPP.c1);
    Y3.c1 = u29_mul4(R.c0, d.c1, R.c1, d.c0, nY0, PPP.c1, nY1, PPP.c0);
    A.zz = f2_mulF29(A.zz, PP, u29_neg<8>(A.zz.c1));
    A.zzz = f2_mulF29(A.zzz, PPP, u29_neg<8>(A.zzz.c1));
    A.x = X3;
    A.y = Y3;
}

// ---- XYZZ + XYZZ and doubling for the bucket-reduction tail (G2), fused multi-product form: every output coordinate is a direct
// product output.  Class invariant proven by tools/u29_model.py (check_add_dbl_class_g2, exact_check_add_dbl_g2): with both
// components of every input coordinate < 32 p and weakly normalised, every output component is again < 32 p (add: <= 20.4 p,
// dbl: <= 22.5 p).  Operation order and bias multiples are exactly add_fp2 / dbl_fp2 of the model.
__device__ __forceinline__ void acc29g2_load(Acc29G2& A, const XYZZ<Fp2>& c) {  // canonical image -> lazily reduced, no multiplication
    A.inf = c.is_inf();
    A.x = f2_load29(c.x);
    A.y = f2_load29(c.y);
    A.zz = f2_load29(c.zz);
    A.zzz = f2_load29(c.zzz);
}
template <int K>
__device__ __forceinline__ U29x2 f2_mulFK29(const U29x2& a, const U29x2& b) {  // fused product, a.c1 negated against K*p
    return f2_mulF29(a, b, u29_neg<K>(a.c1));
}
// x (a single-product output, < 16 p) can only be == 0 mod p if (x mod 2^29) * p^-1 mod 2^29 < 16: a filter with a false-positive
// rate of 2^-25 that never misses
__device__ __forceinline__ bool u29_maybe_zero16(const U29& x) { return (((x.l[0] & Fp29::MASK) * Fp29::PINV) & Fp29::MASK) < 16u; }

__device__ __forceinline__ void acc29g2_dbl(Acc29G2& A) {  // dbl-2008-s-1
    if (A.inf) return;
    const U29 one = u29_one();
    const U29 yc0 = u29_mul(A.y.c0, one), yc1 = u29_mul(A.y.c1, one);  // contracted: keeps U = 2 Y small
    const U29x2 U{u29_wnorm(u29_add(yc0, yc0)), u29_wnorm(u29_add(yc1, yc1))};
    const U29 nU1 = u29_neg<4>(U.c1);
    U29x2 V;
    V.c0 = u29_mul2(U.c0, U.c0, nU1, U.c1);
    V.c1 = u29_mul(u29_add(U.c0, U.c0), U.c1);
    const U29x2 Wv = f2_mulF29(U, V, nU1);
    const U29 nX1 = u29_neg<40>(A.x.c1);
    const U29x2 S = f2_mulF29(A.x, V, nX1);
    U29x2 X2;
    X2.c0 = u29_mul2(A.x.c0, A.x.c0, nX1, A.x.c1);
    X2.c1 = u29_mul(u29_add(A.x.c0, A.x.c0), A.x.c1);
    const U29x2 M{u29_wnorm(u29_add(u29_add(X2.c0, X2.c0), X2.c0)), u29_wnorm(u29_add(u29_add(X2.c1, X2.c1), X2.c1))};
    const U29 nM1 = u29_neg<40>(M.c1);
    const U29 tX0 = u29_wnorm(u29_add(A.x.c0, A.x.c0)), tX1 = u29_wnorm(u29_add(A.x.c1, A.x.c1));
    const U29 n2X0 = u29_neg<80>(tX0), n2X1 = u29_neg<80>(tX1);
    U29x2 X3;  // M^2 - 2 X1 V
    X3.c0 = u29_mul4(M.c0, M.c0, nM1, M.c1, n2X0, V.c0, tX1, V.c1);
    X3.c1 = u29_mul3(u29_add(M.c0, M.c0), M.c1, n2X0, V.c1, n2X1, V.c0);
    const U29x2 d = f2_sub29<24>(S, X3);
    const U29 nW0 = u29_neg<4>(Wv.c0), nW1 = u29_neg<4>(Wv.c1);
    U29x2 Y3;  // M (S - X3) - W Y1
    Y3.c0 = u29_mul4(M.c0, d.c0, nM1, d.c1, nW0, A.y.c0, Wv.c1, A.y.c1);
    Y3.c1 = u29_mul4(M.c0, d.c1, M.c1, d.c0, nW0, A.y.c1, nW1, A.y.c0);
    A.zz = f2_mulFK29<4>(V, A.zz);
    A.zzz = f2_mulF29(Wv, A.zzz, nW1);
    A.x = X3;
    A.y = Y3;
}
__device__ __forceinline__ void acc29g2_add(Acc29G2& A, const Acc29G2& Bq) {  // add-2008-s
    if (Bq.inf) return;
    if (A.inf) { A = Bq; return; }
    const U29x2 U1 = f2_mulFK29<40>(A.x, Bq.zz), U2 = f2_mulFK29<40>(Bq.x, A.zz);
    const U29x2 S1 = f2_mulFK29<40>(A.y, Bq.zzz), S2 = f2_mulFK29<40>(Bq.y, A.zzz);
    const U29x2 P = f2_sub29<16>(U2, U1);
    const U29x2 R = f2_sub29<16>(S2, S1);
    const U29 nP1 = u29_neg<32>(P.c1), nR1 = u29_neg<32>(R.c1);
    U29x2 PP;
    PP.c0 = u29_mul2(P.c0, P.c0, nP1, P.c1);
    PP.c1 = u29_mul(u29_add(P.c0, P.c0), P.c1);
    if (u29_maybe_zero16(PP.c1)) {
        // 2 P0 P1 == 0 mod p is NECESSARY for P == 0 (same x: doubling or P + (-P)); the canonical saturated addition is correct for
        // every input, so it is taken whenever the filter fires
        XYZZ<Fp2> a = acc29g2_to_xyzz(A);
        a.add(acc29g2_to_xyzz(Bq));
        acc29g2_load(A, a);
        return;
    }
    const U29x2 PPP = f2_mulF29(P, PP, nP1);
    const U29x2 Q = f2_mulFK29<16>(U1, PP);
    const U29 W0 = u29_wnorm(u29_add(P.c0, u29_add(U1.c0, U1.c0))), W1 = u29_wnorm(u29_add(P.c1, u29_add(U1.c1, U1.c1)));
    const U29 nW0 = u29_neg<64>(W0), nW1 = u29_neg<64>(W1);
    U29x2 X3;  // R^2 - (P + 2 U1) PP
    X3.c0 = u29_mul4(R.c0, R.c0, nR1, R.c1, nW0, PP.c0, W1, PP.c1);
    X3.c1 = u29_mul3(u29_add(R.c0, R.c0), R.c1, nW0, PP.c1, nW1, PP.c0);
    const U29x2 d = f2_sub29<24>(Q, X3);
    const U29 nS0 = u29_neg<16>(S1.c0), nS1 = u29_neg<16>(S1.c1);
    U29x2 Y3;  // R (Q - X3) - S1 PPP
    Y3.c0 = u29_mul4(R.c0, d.c0, nR1, d.c1, nS0, PPP.c0, S1.c1, PPP.c1);
    Y3.c1 = u29_mul4(R.c0, d.c1, R.c1, d.c0, nS0, PPP.c1, nS1, PPP.c0);
    const U29x2 ZZ = f2_mulFK29<40>(A.zz, Bq.zz);
    const U29x2 ZZZ = f2_mulFK29<40>(A.zzz, Bq.zzz);
    A.zz = f2_mulFK29<16>(ZZ, PP);
    A.zzz = f2_mulFK29<16>(ZZZ, PPP);
    A.x = X3;
    A.y = Y3;
}

// ------------------------------------------------------------------------------------------------------------ Fr (NTT)
// The same 9 x 29-bit representation over the scalar field, for the NTT butterflies (ntt.hip).  Data stay congruent to gnark's
// Montgomery-2^256 image (unpacked WITHOUT a shift); twiddles are kept as w * 2^261 mod r in a second table, so that
// mont29(X, W') = X * W' / 2^261 is the exact field product in the 2^256 domain and a pass ends with a partial reduction and a
// repacking instead of a multiplication.  tools/u29_ntt_model.py proves the bounds of the stage-group schedules of ntt.hip (no
// 64-bit column / 32-bit limb overflow, every bias dominates its subtrahend) and checks whole transforms against the oracle.
struct Fr29 {
    static constexpr uint32_t MASK = 0x1fffffffu;
    static constexpr uint32_t P[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    static constexpr uint32_t NINV = 0x0fffffffu;  // -r^-1 mod 2^29
    static constexpr uint32_t RC[9] = {0x0fffffffu, 0x00f05360u, 0x11a3dbafu, 0x182f6f0cu, 0x0a7a2d7cu, 0x1d24bf3fu, 0x1f591ebeu, 0x11a3d9cbu, 0x1fcf9bb1u};  // 2^261 - r
    static constexpr uint32_t QM = 0xa948e6d7u;    // floor(2^53 / ((r >> 232) + 1))
    static constexpr uint32_t BIAS4[9] = {0x40000004u, 0x5c3eb27cu, 0x59709141u, 0x5f4243cbu, 0x56174a0au, 0x4b6d0300u, 0x429b8502u, 0x597098ceu, 0x00c19137u};
    static constexpr uint32_t BIAS16[9] = {0x40000010u, 0x50fac9f6u, 0x45c2450du, 0x5d090f35u, 0x585d2831u, 0x4db40c08u, 0x4a6e140fu, 0x45c2633eu, 0x030644e5u};
    static constexpr uint32_t BIAS24[9] = {0x40000018u, 0x49782ef2u, 0x58a36795u, 0x5b8d96d0u, 0x448bbc4bu, 0x448e120eu, 0x4fa51e18u, 0x58a394deu, 0x04896758u};
    static constexpr uint32_t BIAS40[9] = {0x40000028u, 0x5a72f8eau, 0x5e65aca4u, 0x5896a607u, 0x5ce8e47fu, 0x52421e18u, 0x5a133229u, 0x5e65f81eu, 0x078fac3fu};
    template <int K>
    static constexpr uint32_t bias(int i) {
        static_assert(K == 4 || K == 16 || K == 24 || K == 40, "no bias table for this multiple of r");
        return K == 4 ? BIAS4[i] : K == 16 ? BIAS16[i] : K == 24 ? BIAS24[i] : BIAS40[i];
    }
};

// a * b / 2^261 mod r (lazily reduced: < a*b/2^261 + r), limbs 0..7 < 2^29
__device__ __forceinline__ U29 u29r_mul(const U29& a, const U29& b) {
    U29 r;
    asm(ZKMI_MONT_MUL29_ASM
        : [r0] "=&v"(r.l[0]), [r1] "=&v"(r.l[1]), [r2] "=&v"(r.l[2]), [r3] "=&v"(r.l[3]), [r4] "=&v"(r.l[4]), [r5] "=&v"(r.l[5]),
          [r6] "=&v"(r.l[6]), [r7] "=&v"(r.l[7]), [r8] "=&v"(r.l[8])
        : [a0] "v"(a.l[0]), [a1] "v"(a.l[1]), [a2] "v"(a.l[2]), [a3] "v"(a.l[3]), [a4] "v"(a.l[4]), [a5] "v"(a.l[5]), [a6] "v"(a.l[6]),
          [a7] "v"(a.l[7]), [a8] "v"(a.l[8]), [b0] "v"(b.l[0]), [b1] "v"(b.l[1]), [b2] "v"(b.l[2]), [b3] "v"(b.l[3]), [b4] "v"(b.l[4]),
          [b5] "v"(b.l[5]), [b6] "v"(b.l[6]), [b7] "v"(b.l[7]), [b8] "v"(b.l[8]), [p0] "s"(Fr29::P[0]), [p1] "s"(Fr29::P[1]),
          [p2] "s"(Fr29::P[2]), [p3] "s"(Fr29::P[3]), [p4] "s"(Fr29::P[4]), [p5] "s"(Fr29::P[5]), [p6] "s"(Fr29::P[6]), [p7] "s"(Fr29::P[7]),
          [p8] "s"(Fr29::P[8]), [ninv] "s"(Fr29::NINV)
        : "v0", "v1", "vcc");
    return r;
}

// a - b + K*r  (b weakly normalised and < K*r with the model's margin)
template <int K>
__device__ __forceinline__ U29 u29r_sub(const U29& a, const U29& b) {
    U29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] - b.l[i] + Fr29::bias<K>(i);
    return r;
}

// x -= q r with q estimated from the top limb (never too large): result < 2.01 r, limbs 0..7 < 2^29.  x weakly normalised.
__device__ __forceinline__ U29 u29r_reduce(const U29& x) {
    const uint32_t q = __umulhi(x.l[8], Fr29::QM) >> 21;
    U29 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t acc = (uint64_t)q * Fr29::RC[i] + (uint64_t)(x.l[i] + c);  // x + q (2^261 - r); the q 2^261 leaves through the top limb
        r.l[i] = (uint32_t)acc & Fr29::MASK;
        c = (uint32_t)(acc >> 29);
    }
    uint64_t acc = (uint64_t)q * Fr29::RC[8] + (uint64_t)(x.l[8] + c);
    r.l[8] = (uint32_t)(acc - ((uint64_t)q << 29));
    return r;
}

// canonical image V -> V << 5 (the multiplier form of a scaling-table entry kept in the 2^256 domain: V << 5 == t * 2^261)
__device__ __forceinline__ U29 u29r_load5(const Fr& v) {
    U29 r;
    const uint32_t* w = v.l;
    r.l[0] = (w[0] << 5) & 0x1fffffffu;
    r.l[1] = __funnelshift_r(w[0], w[1], 24) & 0x1fffffffu;
    r.l[2] = __funnelshift_r(w[1], w[2], 21) & 0x1fffffffu;
    r.l[3] = __funnelshift_r(w[2], w[3], 18) & 0x1fffffffu;
    r.l[4] = __funnelshift_r(w[3], w[4], 15) & 0x1fffffffu;
    r.l[5] = __funnelshift_r(w[4], w[5], 12) & 0x1fffffffu;
    r.l[6] = __funnelshift_r(w[5], w[6], 9) & 0x1fffffffu;
    r.l[7] = __funnelshift_r(w[6], w[7], 6) & 0x1fffffffu;
    r.l[8] = w[7] >> 3;
    return r;
}

// x as left by u29r_reduce (< 2.01 r, limbs 0..7 < 2^29) -> 8 x u32; canonical: two conditional subtractions of r
__device__ __forceinline__ Fr u29r_pack(const U29& t, bool canonical) {
    Fr r;
    r.l[0] = t.l[0] | (t.l[1] << 29);
    r.l[1] = (t.l[1] >> 3) | (t.l[2] << 26);
    r.l[2] = (t.l[2] >> 6) | (t.l[3] << 23);
    r.l[3] = (t.l[3] >> 9) | (t.l[4] << 20);
    r.l[4] = (t.l[4] >> 12) | (t.l[5] << 17);
    r.l[5] = (t.l[5] >> 15) | (t.l[6] << 14);
    r.l[6] = (t.l[6] >> 18) | (t.l[7] << 11);
    r.l[7] = (t.l[7] >> 21) | (t.l[8] << 8);
    if (canonical) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            uint32_t s[8];
            uint64_t bw = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                uint64_t d = (uint64_t)r.l[i] - FrParams::MOD[i] - bw;
                s[i] = (uint32_t)d;
                bw = d >> 63;
            }
            if (!bw) {
#pragma unroll
                for (int i = 0; i < 8; i++) r.l[i] = s[i];
            }
        }
    }
    return r;
}

}  // namespace zkmi
