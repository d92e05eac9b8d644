// BN254 prime-field arithmetic for gfx950 (CDNA4) and for the host-side O(1) finishing steps.
//
// Replaces (on the device) the 4x64-bit Montgomery arithmetic of gnark-crypto v0.9.1 `ecc/bn254/fr` and `ecc/bn254/fp`
// (pinned at /root/reference/gnark_backend_ffi/go.mod:5; used as `fr_bn254.Element` at
// /root/reference/gnark_backend_ffi/main.go:16,81-82 and acir/term/mul_term.go:16).  Memory image is identical:
// 32 bytes, little-endian, Montgomery form x*2^256 mod p -- so 8 x u32 limbs here alias Go's [4]uint64.
//
// Device multiply = product-scanning (FIPS) Montgomery on 32-bit limbs: every 32x32->64 product is ONE
// v_mad_u64_u32 accumulating into a 96-bit column accumulator (carry-out -> v_addc).  CDNA4 has no 64-bit integer
// multiplier; v_mad_u64_u32 is the widest multiply the VALU offers.  No MFMA: there is no dense contraction here.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ff_mul_gfx950.inc"

#define ZK_HD __host__ __device__ __forceinline__
#define ZK_D __device__ __forceinline__

namespace zkmi {

struct FrParams {
    static constexpr uint32_t MOD[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
    static constexpr uint32_t NINV = 0xefffffffu;  // -r^-1 mod 2^32 (low half of gnark's qInvNeg)
};
struct FpParams {
    static constexpr uint32_t MOD[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t ONE[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
    static constexpr uint32_t NINV = 0xe4866389u;
};

// ---- 96-bit column accumulator -------------------------------------------------------------------------------
struct Acc96 {
    uint64_t lo;  // acc1:acc0
    uint32_t hi;  // acc2
};

// acc += a*b   (host path only; the device path is the generated asm schedule in ff_mul_gfx950.inc)
ZK_HD void mac(Acc96& c, uint32_t a, uint32_t b) {
    uint64_t p = (uint64_t)a * b;
    uint64_t s = c.lo + p;
    c.hi += (s < p);
    c.lo = s;
}
ZK_HD void shift32(Acc96& c) {
    c.lo = (c.lo >> 32) | ((uint64_t)c.hi << 32);
    c.hi = 0;
}

template <class P>
struct Field {
    uint32_t l[8];

    static ZK_HD Field zero() { Field r; for (int i = 0; i < 8; i++) r.l[i] = 0; return r; }
    static ZK_HD Field one() { Field r; for (int i = 0; i < 8; i++) r.l[i] = P::ONE[i]; return r; }
    static ZK_HD Field r2() { Field r; for (int i = 0; i < 8; i++) r.l[i] = P::R2[i]; return r; }
    static ZK_HD Field modulus() { Field r; for (int i = 0; i < 8; i++) r.l[i] = P::MOD[i]; return r; }

    ZK_HD bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= l[i];
        return o == 0;
    }
    ZK_HD bool operator==(const Field& b) const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= l[i] ^ b.l[i];
        return o == 0;
    }
    ZK_HD bool operator!=(const Field& b) const { return !(*this == b); }

    // r = t - MOD if t >= MOD (t < 2*MOD)
    static ZK_HD Field reduce_once(const uint32_t t[8]) {
        Field r;
#if defined(__HIP_DEVICE_COMPILE__)
        // borrow chain t - MOD, then select by the final borrow.  MOD limbs as VGPR operands: a VOP2 carry-in
        // (implicit VCC read) plus an SGPR source would exceed the gfx9 constant-bus limit.
        asm("v_sub_co_u32_e32 %[s0], vcc, %[t0], %[p0]\n\t"
            "v_subb_co_u32_e32 %[s1], vcc, %[t1], %[p1], vcc\n\t"
            "v_subb_co_u32_e32 %[s2], vcc, %[t2], %[p2], vcc\n\t"
            "v_subb_co_u32_e32 %[s3], vcc, %[t3], %[p3], vcc\n\t"
            "v_subb_co_u32_e32 %[s4], vcc, %[t4], %[p4], vcc\n\t"
            "v_subb_co_u32_e32 %[s5], vcc, %[t5], %[p5], vcc\n\t"
            "v_subb_co_u32_e32 %[s6], vcc, %[t6], %[p6], vcc\n\t"
            "v_subb_co_u32_e32 %[s7], vcc, %[t7], %[p7], vcc\n\t"
            "v_cndmask_b32_e32 %[s0], %[s0], %[t0], vcc\n\t"
            "v_cndmask_b32_e32 %[s1], %[s1], %[t1], vcc\n\t"
            "v_cndmask_b32_e32 %[s2], %[s2], %[t2], vcc\n\t"
            "v_cndmask_b32_e32 %[s3], %[s3], %[t3], vcc\n\t"
            "v_cndmask_b32_e32 %[s4], %[s4], %[t4], vcc\n\t"
            "v_cndmask_b32_e32 %[s5], %[s5], %[t5], vcc\n\t"
            "v_cndmask_b32_e32 %[s6], %[s6], %[t6], vcc\n\t"
            "v_cndmask_b32_e32 %[s7], %[s7], %[t7], vcc\n\t"
            : [s0] "=&v"(r.l[0]), [s1] "=&v"(r.l[1]), [s2] "=&v"(r.l[2]), [s3] "=&v"(r.l[3]), [s4] "=&v"(r.l[4]),
              [s5] "=&v"(r.l[5]), [s6] "=&v"(r.l[6]), [s7] "=&v"(r.l[7])
            : [t0] "v"(t[0]), [t1] "v"(t[1]), [t2] "v"(t[2]), [t3] "v"(t[3]), [t4] "v"(t[4]), [t5] "v"(t[5]), [t6] "v"(t[6]),
              [t7] "v"(t[7]), [p0] "v"(P::MOD[0]), [p1] "v"(P::MOD[1]), [p2] "v"(P::MOD[2]), [p3] "v"(P::MOD[3]),
              [p4] "v"(P::MOD[4]), [p5] "v"(P::MOD[5]), [p6] "v"(P::MOD[6]), [p7] "v"(P::MOD[7])
            : "vcc");
#else
        uint32_t s[8];
        uint64_t bw = 0;
        for (int i = 0; i < 8; i++) {
            uint64_t d = (uint64_t)t[i] - P::MOD[i] - bw;
            s[i] = (uint32_t)d;
            bw = (d >> 63);
        }
        for (int i = 0; i < 8; i++) r.l[i] = bw ? t[i] : s[i];
#endif
        return r;
    }

    friend ZK_HD Field operator+(const Field& a, const Field& b) {
        uint32_t t[8];
#if defined(__HIP_DEVICE_COMPILE__)
        asm("v_add_co_u32_e32 %[t0], vcc, %[a0], %[b0]\n\t"
            "v_addc_co_u32_e32 %[t1], vcc, %[a1], %[b1], vcc\n\t"
            "v_addc_co_u32_e32 %[t2], vcc, %[a2], %[b2], vcc\n\t"
            "v_addc_co_u32_e32 %[t3], vcc, %[a3], %[b3], vcc\n\t"
            "v_addc_co_u32_e32 %[t4], vcc, %[a4], %[b4], vcc\n\t"
            "v_addc_co_u32_e32 %[t5], vcc, %[a5], %[b5], vcc\n\t"
            "v_addc_co_u32_e32 %[t6], vcc, %[a6], %[b6], vcc\n\t"
            "v_addc_co_u32_e32 %[t7], vcc, %[a7], %[b7], vcc\n\t"
            : [t0] "=&v"(t[0]), [t1] "=&v"(t[1]), [t2] "=&v"(t[2]), [t3] "=&v"(t[3]), [t4] "=&v"(t[4]), [t5] "=&v"(t[5]),
              [t6] "=&v"(t[6]), [t7] "=&v"(t[7])
            : [a0] "v"(a.l[0]), [a1] "v"(a.l[1]), [a2] "v"(a.l[2]), [a3] "v"(a.l[3]), [a4] "v"(a.l[4]), [a5] "v"(a.l[5]),
              [a6] "v"(a.l[6]), [a7] "v"(a.l[7]), [b0] "v"(b.l[0]), [b1] "v"(b.l[1]), [b2] "v"(b.l[2]), [b3] "v"(b.l[3]),
              [b4] "v"(b.l[4]), [b5] "v"(b.l[5]), [b6] "v"(b.l[6]), [b7] "v"(b.l[7])
            : "vcc");
#else
        uint64_t c = 0;
        for (int i = 0; i < 8; i++) {
            c += (uint64_t)a.l[i] + b.l[i];
            t[i] = (uint32_t)c;
            c >>= 32;
        }
#endif
        return reduce_once(t);  // MOD < 2^254: no carry out of the top limb
    }
    friend ZK_HD Field operator-(const Field& a, const Field& b) {
        Field r;
#if defined(__HIP_DEVICE_COMPILE__)
        uint32_t t[8], m[8];
        // a - b, then add back (borrow ? MOD : 0)
        asm("v_sub_co_u32_e32 %[t0], vcc, %[a0], %[b0]\n\t"
            "v_subb_co_u32_e32 %[t1], vcc, %[a1], %[b1], vcc\n\t"
            "v_subb_co_u32_e32 %[t2], vcc, %[a2], %[b2], vcc\n\t"
            "v_subb_co_u32_e32 %[t3], vcc, %[a3], %[b3], vcc\n\t"
            "v_subb_co_u32_e32 %[t4], vcc, %[a4], %[b4], vcc\n\t"
            "v_subb_co_u32_e32 %[t5], vcc, %[a5], %[b5], vcc\n\t"
            "v_subb_co_u32_e32 %[t6], vcc, %[a6], %[b6], vcc\n\t"
            "v_subb_co_u32_e32 %[t7], vcc, %[a7], %[b7], vcc\n\t"
            "v_cndmask_b32_e32 %[m0], 0, %[p0], vcc\n\t"
            "v_cndmask_b32_e32 %[m1], 0, %[p1], vcc\n\t"
            "v_cndmask_b32_e32 %[m2], 0, %[p2], vcc\n\t"
            "v_cndmask_b32_e32 %[m3], 0, %[p3], vcc\n\t"
            "v_cndmask_b32_e32 %[m4], 0, %[p4], vcc\n\t"
            "v_cndmask_b32_e32 %[m5], 0, %[p5], vcc\n\t"
            "v_cndmask_b32_e32 %[m6], 0, %[p6], vcc\n\t"
            "v_cndmask_b32_e32 %[m7], 0, %[p7], vcc\n\t"
            "v_add_co_u32_e32 %[t0], vcc, %[t0], %[m0]\n\t"
            "v_addc_co_u32_e32 %[t1], vcc, %[t1], %[m1], vcc\n\t"
            "v_addc_co_u32_e32 %[t2], vcc, %[t2], %[m2], vcc\n\t"
            "v_addc_co_u32_e32 %[t3], vcc, %[t3], %[m3], vcc\n\t"
            "v_addc_co_u32_e32 %[t4], vcc, %[t4], %[m4], vcc\n\t"
            "v_addc_co_u32_e32 %[t5], vcc, %[t5], %[m5], vcc\n\t"
            "v_addc_co_u32_e32 %[t6], vcc, %[t6], %[m6], vcc\n\t"
            "v_addc_co_u32_e32 %[t7], vcc, %[t7], %[m7], vcc\n\t"
            : [t0] "=&v"(t[0]), [t1] "=&v"(t[1]), [t2] "=&v"(t[2]), [t3] "=&v"(t[3]), [t4] "=&v"(t[4]), [t5] "=&v"(t[5]),
              [t6] "=&v"(t[6]), [t7] "=&v"(t[7]), [m0] "=&v"(m[0]), [m1] "=&v"(m[1]), [m2] "=&v"(m[2]), [m3] "=&v"(m[3]),
              [m4] "=&v"(m[4]), [m5] "=&v"(m[5]), [m6] "=&v"(m[6]), [m7] "=&v"(m[7])
            : [a0] "v"(a.l[0]), [a1] "v"(a.l[1]), [a2] "v"(a.l[2]), [a3] "v"(a.l[3]), [a4] "v"(a.l[4]), [a5] "v"(a.l[5]),
              [a6] "v"(a.l[6]), [a7] "v"(a.l[7]), [b0] "v"(b.l[0]), [b1] "v"(b.l[1]), [b2] "v"(b.l[2]), [b3] "v"(b.l[3]),
              [b4] "v"(b.l[4]), [b5] "v"(b.l[5]), [b6] "v"(b.l[6]), [b7] "v"(b.l[7]), [p0] "v"(P::MOD[0]), [p1] "v"(P::MOD[1]),
              [p2] "v"(P::MOD[2]), [p3] "v"(P::MOD[3]), [p4] "v"(P::MOD[4]), [p5] "v"(P::MOD[5]), [p6] "v"(P::MOD[6]),
              [p7] "v"(P::MOD[7])
            : "vcc");
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = t[i];
#else
        uint32_t t[8];
        uint64_t bw = 0;
        for (int i = 0; i < 8; i++) {
            uint64_t d = (uint64_t)a.l[i] - b.l[i] - bw;
            t[i] = (uint32_t)d;
            bw = (d >> 63);
        }
        uint32_t mask = (uint32_t)0 - (uint32_t)bw;
        uint64_t c = 0;
        for (int i = 0; i < 8; i++) {
            c += (uint64_t)t[i] + (P::MOD[i] & mask);
            r.l[i] = (uint32_t)c;
            c >>= 32;
        }
#endif
        return r;
    }
    ZK_HD Field neg() const { return is_zero() ? *this : (modulus_raw_sub(*this)); }
    static ZK_HD Field modulus_raw_sub(const Field& a) {
        Field r;
        uint64_t bw = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t d = (uint64_t)P::MOD[i] - a.l[i] - bw;
            r.l[i] = (uint32_t)d;
            bw = (d >> 63);
        }
        return r;
    }
    ZK_HD Field dbl() const { return *this + *this; }

    // Montgomery product a*b*2^-256 mod p  (FIPS / product scanning)
    friend ZK_HD Field operator*(const Field& a, const Field& b) {
        uint32_t t[8];
#if defined(__HIP_DEVICE_COMPILE__)
        // one asm statement = the whole 128-product schedule (gen_ff_asm.py); accumulator in v[0:3]
        asm(ZKMI_MONT_MUL_ASM
            : [r0] "=&v"(t[0]), [r1] "=&v"(t[1]), [r2] "=&v"(t[2]), [r3] "=&v"(t[3]), [r4] "=&v"(t[4]), [r5] "=&v"(t[5]),
              [r6] "=&v"(t[6]), [r7] "=&v"(t[7])
            : [a0] "v"(a.l[0]), [a1] "v"(a.l[1]), [a2] "v"(a.l[2]), [a3] "v"(a.l[3]), [a4] "v"(a.l[4]), [a5] "v"(a.l[5]),
              [a6] "v"(a.l[6]), [a7] "v"(a.l[7]), [b0] "v"(b.l[0]), [b1] "v"(b.l[1]), [b2] "v"(b.l[2]), [b3] "v"(b.l[3]),
              [b4] "v"(b.l[4]), [b5] "v"(b.l[5]), [b6] "v"(b.l[6]), [b7] "v"(b.l[7]), [p0] "s"(P::MOD[0]), [p1] "s"(P::MOD[1]),
              [p2] "s"(P::MOD[2]), [p3] "s"(P::MOD[3]), [p4] "s"(P::MOD[4]), [p5] "s"(P::MOD[5]), [p6] "s"(P::MOD[6]),
              [p7] "s"(P::MOD[7]), [ninv] "s"(P::NINV)
            : "v0", "v1", "v2", "v3", "vcc");
#else
        Acc96 c{0, 0};
        uint32_t m[8];
        for (int k = 0; k < 8; k++) {
            for (int j = 0; j < k; j++) {
                mac(c, a.l[j], b.l[k - j]);
                mac(c, m[j], P::MOD[k - j]);
            }
            mac(c, a.l[k], b.l[0]);
            m[k] = (uint32_t)c.lo * P::NINV;
            mac(c, m[k], P::MOD[0]);
            shift32(c);
        }
        for (int k = 8; k < 16; k++) {
            for (int j = k - 7; j < 8; j++) {
                mac(c, a.l[j], b.l[k - j]);
                mac(c, m[j], P::MOD[k - j]);
            }
            t[k - 8] = (uint32_t)c.lo;
            shift32(c);
        }
#endif
        return reduce_once(t);
    }
    ZK_HD Field sqr() const { return *this * *this; }

    ZK_HD Field to_mont() const { return *this * r2(); }
    ZK_HD Field from_mont() const {
        Field o = zero();
        o.l[0] = 1;
        return *this * o;
    }
    // a^e, e = 8 x u32 little-endian (not constant time; only public data flows through here)
    ZK_HD Field pow(const uint32_t e[8]) const {
        Field acc = one(), base = *this;
        for (int i = 0; i < 256; i++) {
            if ((e[i >> 5] >> (i & 31)) & 1) acc = acc * base;
            base = base.sqr();
        }
        return acc;
    }
    // Fermat inverse; inv(0) = 0 (gnark-crypto Element.Inverse convention)
    ZK_HD Field inv() const {
        uint32_t e[8];
        for (int i = 0; i < 8; i++) e[i] = P::MOD[i];
        e[0] -= 2;  // MOD[0] >= 2 for both fields
        return pow(e);
    }
    static ZK_HD Field from_u32(uint32_t v) {
        Field o = zero();
        o.l[0] = v;
        return o.to_mont();
    }
};

using Fr = Field<FrParams>;
using Fp = Field<FpParams>;

// ---- Fp2 = Fp[u]/(u^2+1)  (gnark-crypto E2{A0, A1}) --------------------------------------------------------------
struct Fp2 {
    Fp a0, a1;
    static ZK_HD Fp2 zero() { return Fp2{Fp::zero(), Fp::zero()}; }
    static ZK_HD Fp2 one() { return Fp2{Fp::one(), Fp::zero()}; }
    ZK_HD bool is_zero() const { return a0.is_zero() && a1.is_zero(); }
    ZK_HD bool operator==(const Fp2& b) const { return a0 == b.a0 && a1 == b.a1; }
    ZK_HD bool operator!=(const Fp2& b) const { return !(*this == b); }
    friend ZK_HD Fp2 operator+(const Fp2& a, const Fp2& b) { return Fp2{a.a0 + b.a0, a.a1 + b.a1}; }
    friend ZK_HD Fp2 operator-(const Fp2& a, const Fp2& b) { return Fp2{a.a0 - b.a0, a.a1 - b.a1}; }
    // Karatsuba: 3 Fp products
    friend ZK_HD Fp2 operator*(const Fp2& a, const Fp2& b) {
        Fp v0 = a.a0 * b.a0, v1 = a.a1 * b.a1;
        Fp s = (a.a0 + a.a1) * (b.a0 + b.a1);
        return Fp2{v0 - v1, s - v0 - v1};
    }
    // (a0+a1)(a0-a1), 2 a0 a1 : 2 Fp products
    ZK_HD Fp2 sqr() const {
        Fp p = a0 * a1;
        return Fp2{(a0 + a1) * (a0 - a1), p + p};
    }
    ZK_HD Fp2 neg() const { return Fp2{a0.neg(), a1.neg()}; }
    ZK_HD Fp2 dbl() const { return Fp2{a0.dbl(), a1.dbl()}; }
    ZK_HD Fp2 inv() const {
        Fp d = (a0.sqr() + a1.sqr()).inv();
        return Fp2{a0 * d, (a1 * d).neg()};
    }
};

}  // namespace zkmi
