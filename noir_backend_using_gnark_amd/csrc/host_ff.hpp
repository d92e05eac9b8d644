// Host-side BN254 Fp / Fp2 arithmetic (4 x u64 Montgomery limbs) for the O(1) finishing steps that stay on the CPU:
// Horner combine of the <= 32 window sums an MSM returns, XYZZ -> affine (one inversion), and the handful of scalar
// multiplications of groth16.Prove (r*delta, s*delta, s*Ar, r*Bs1, rs*delta) -- gnark does these on the CPU too
// (gnark v0.8.0 groth16 prove.go, reached from /root/reference/gnark_backend_ffi/main.go:131).  Same 32-byte memory
// image as the device type zkmi::Field (ff.hpp), so values move between the two with memcpy.
// This is product code (not the oracle under oracle/): it implements the same interface as zkmi::Field so that the
// curve templates in curve.hpp instantiate over it.
#pragma once
#include <stdint.h>
#include <string.h>

namespace zkmi {

struct HFpParams {
    static constexpr uint64_t MOD[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    static constexpr uint64_t ONE[4] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL};
    static constexpr uint64_t R2[4] = {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL};
    static constexpr uint64_t NINV = 0x87d20782e4866389ULL;
};
struct HFrParams {
    static constexpr uint64_t MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    static constexpr uint64_t ONE[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
    static constexpr uint64_t R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};
    static constexpr uint64_t NINV = 0xc2e1f593efffffffULL;
};

template <class P>
struct HField {
    uint64_t l[4];
    typedef unsigned __int128 u128;

    static HField zero() { return HField{{0, 0, 0, 0}}; }
    static HField one() { return HField{{P::ONE[0], P::ONE[1], P::ONE[2], P::ONE[3]}}; }
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
    bool operator==(const HField& b) const { return l[0] == b.l[0] && l[1] == b.l[1] && l[2] == b.l[2] && l[3] == b.l[3]; }
    bool operator!=(const HField& b) const { return !(*this == b); }

    static bool geq_mod(const uint64_t t[4]) {
        for (int i = 3; i >= 0; i--)
            if (t[i] != P::MOD[i]) return t[i] > P::MOD[i];
        return true;
    }
    static void sub_mod(uint64_t t[4]) {
        uint64_t bw = 0;
        for (int i = 0; i < 4; i++) {
            u128 d = (u128)t[i] - P::MOD[i] - bw;
            t[i] = (uint64_t)d;
            bw = (uint64_t)(d >> 64) & 1;
        }
    }
    friend HField operator+(const HField& a, const HField& b) {
        HField r;
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)a.l[i] + b.l[i];
            r.l[i] = (uint64_t)c;
            c >>= 64;
        }
        if (geq_mod(r.l)) sub_mod(r.l);
        return r;
    }
    friend HField operator-(const HField& a, const HField& b) {
        HField r;
        uint64_t bw = 0;
        for (int i = 0; i < 4; i++) {
            u128 d = (u128)a.l[i] - b.l[i] - bw;
            r.l[i] = (uint64_t)d;
            bw = (uint64_t)(d >> 64) & 1;
        }
        if (bw) {
            u128 c = 0;
            for (int i = 0; i < 4; i++) {
                c += (u128)r.l[i] + P::MOD[i];
                r.l[i] = (uint64_t)c;
                c >>= 64;
            }
        }
        return r;
    }
    HField neg() const { return is_zero() ? *this : (zero() - *this); }
    HField dbl() const { return *this + *this; }
    // coarsely-integrated operand scanning, 64-bit words
    friend HField operator*(const HField& a, const HField& b) {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) {
                c += (u128)a.l[j] * b.l[i] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[4] = (uint64_t)c;
            t[5] = (uint64_t)(c >> 64);
            uint64_t m = t[0] * P::NINV;
            c = ((u128)m * P::MOD[0] + t[0]) >> 64;
            for (int j = 1; j < 4; j++) {
                c += (u128)m * P::MOD[j] + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[3] = (uint64_t)c;
            t[4] = t[5] + (uint64_t)(c >> 64);
        }
        HField r{{t[0], t[1], t[2], t[3]}};
        if (t[4] || geq_mod(r.l)) sub_mod(r.l);
        return r;
    }
    HField sqr() const { return *this * *this; }
    HField pow(const uint64_t e[4]) const {
        HField acc = one(), base = *this;
        for (int i = 0; i < 256; i++) {
            if ((e[i >> 6] >> (i & 63)) & 1) acc = acc * base;
            base = base.sqr();
        }
        return acc;
    }
    HField inv() const {
        uint64_t e[4] = {P::MOD[0] - 2, P::MOD[1], P::MOD[2], P::MOD[3]};
        return pow(e);
    }
    HField to_mont() const { return *this * HField{{P::R2[0], P::R2[1], P::R2[2], P::R2[3]}}; }
    HField from_mont() const { return *this * HField{{1, 0, 0, 0}}; }
};

using HFp = HField<HFpParams>;
using HFr = HField<HFrParams>;

struct HFp2 {
    HFp a0, a1;
    static HFp2 zero() { return HFp2{HFp::zero(), HFp::zero()}; }
    static HFp2 one() { return HFp2{HFp::one(), HFp::zero()}; }
    bool is_zero() const { return a0.is_zero() && a1.is_zero(); }
    bool operator==(const HFp2& b) const { return a0 == b.a0 && a1 == b.a1; }
    bool operator!=(const HFp2& b) const { return !(*this == b); }
    friend HFp2 operator+(const HFp2& a, const HFp2& b) { return HFp2{a.a0 + b.a0, a.a1 + b.a1}; }
    friend HFp2 operator-(const HFp2& a, const HFp2& b) { return HFp2{a.a0 - b.a0, a.a1 - b.a1}; }
    friend HFp2 operator*(const HFp2& a, const HFp2& b) {
        HFp v0 = a.a0 * b.a0, v1 = a.a1 * b.a1;
        HFp s = (a.a0 + a.a1) * (b.a0 + b.a1);
        return HFp2{v0 - v1, s - v0 - v1};
    }
    HFp2 sqr() const {
        HFp p = a0 * a1;
        return HFp2{(a0 + a1) * (a0 - a1), p + p};
    }
    HFp2 neg() const { return HFp2{a0.neg(), a1.neg()}; }
    HFp2 dbl() const { return HFp2{a0.dbl(), a1.dbl()}; }
    HFp2 inv() const {
        HFp d = (a0.sqr() + a1.sqr()).inv();
        return HFp2{a0 * d, (a1 * d).neg()};
    }
};

}  // namespace zkmi
