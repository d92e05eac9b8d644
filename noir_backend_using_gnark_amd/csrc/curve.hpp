// BN254 G1 / G2 group arithmetic (host + device), extended-Jacobian "XYZZ" coordinates.
//
// Replaces gnark-crypto v0.9.1 `ecc/bn254` G1Affine/G2Affine + g1JacExtended/g2JacExtended bucket arithmetic used by
// (*G1Jac).MultiExp / (*G2Jac).MultiExp (pinned at /root/reference/gnark_backend_ffi/go.mod:5; reached through
// groth16.Prove at /root/reference/gnark_backend_ffi/main.go:131 and plonk.Prove at backend/plonk/plonk.go:67).
// Memory images match Go's: G1Affine = {X, Y fp.Element} 64 B; G2Affine = {X{A0,A1}, Y{A0,A1}} 128 B; infinity = (0,0).
// Formulas: EFD short-Weierstrass a=0 xyzz (madd-2008-s, mdbl-2008-s-1, add-2008-s, dbl-2008-s-1).
#pragma once
#include "ff.hpp"

namespace zkmi {

template <class F>
struct Affine {
    F x, y;
    ZK_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    static ZK_HD Affine inf() { return Affine{F::zero(), F::zero()}; }
    ZK_HD Affine neg() const { return Affine{x, y.neg()}; }
};

template <class F>
struct XYZZ {
    F x, y, zz, zzz;
    ZK_HD bool is_inf() const { return zz.is_zero(); }
    static ZK_HD XYZZ inf() { return XYZZ{F::zero(), F::zero(), F::zero(), F::zero()}; }
    static ZK_HD XYZZ from_affine(const Affine<F>& p) {
        if (p.is_inf()) return inf();
        return XYZZ{p.x, p.y, F::one(), F::one()};
    }

    // this = 2*p (affine p)
    static ZK_HD XYZZ dbl_affine(const Affine<F>& p) {
        if (p.is_inf()) return inf();
        F u = p.y.dbl();
        F v = u.sqr();
        F w = u * v;
        F s = p.x * v;
        F m = p.x.sqr();
        m = m.dbl() + m;
        XYZZ r;
        r.x = m.sqr() - s - s;
        r.y = m * (s - r.x) - w * p.y;
        r.zz = v;
        r.zzz = w;
        return r;
    }
    ZK_HD void dbl() {
        if (is_inf()) return;
        F u = y.dbl();
        F v = u.sqr();
        F w = u * v;
        F s = x * v;
        F m = x.sqr();
        m = m.dbl() + m;
        F x3 = m.sqr() - s - s;
        y = m * (s - x3) - w * y;
        x = x3;
        zz = v * zz;
        zzz = w * zzz;
    }
    // this += (px, py) affine (already sign-adjusted by the caller); full special-case handling
    ZK_HD void madd(const F& px, const F& py) {
        if (px.is_zero() && py.is_zero()) return;
        if (is_inf()) {
            x = px; y = py; zz = F::one(); zzz = F::one();
            return;
        }
        F pp = px * zz - x;    // P
        F rr = py * zzz - y;   // R
        if (pp.is_zero()) {
            if (rr.is_zero()) *this = dbl_affine(Affine<F>{px, py});
            else *this = inf();
            return;
        }
        F p2 = pp.sqr();
        F ppp = pp * p2;
        F q = x * p2;
        F x3 = rr.sqr() - ppp - q - q;
        y = rr * (q - x3) - y * ppp;
        x = x3;
        zz = zz * p2;
        zzz = zzz * ppp;
    }
    ZK_HD void madd(const Affine<F>& p) { madd(p.x, p.y); }
    ZK_HD void msub(const Affine<F>& p) { madd(p.x, p.y.neg()); }

    // this += o
    ZK_HD void add(const XYZZ& o) {
        if (o.is_inf()) return;
        if (is_inf()) { *this = o; return; }
        F u1 = x * o.zz, u2 = o.x * zz;
        F s1 = y * o.zzz, s2 = o.y * zzz;
        F pp = u2 - u1, rr = s2 - s1;
        if (pp.is_zero()) {
            if (rr.is_zero()) dbl();
            else *this = inf();
            return;
        }
        F p2 = pp.sqr();
        F ppp = pp * p2;
        F q = u1 * p2;
        F x3 = rr.sqr() - ppp - q - q;
        y = rr * (q - x3) - s1 * ppp;
        x = x3;
        zz = zz * o.zz * p2;
        zzz = zzz * o.zzz * ppp;
    }
    ZK_HD XYZZ neg() const { return XYZZ{x, y.neg(), zz, zzz}; }

    ZK_HD Affine<F> to_affine() const {
        if (is_inf()) return Affine<F>::inf();
        F zi = (zz * zzz).inv();
        F izz = zi * zzz, izzz = zi * zz;
        return Affine<F>{x * izz, y * izzz};
    }
};

using G1Affine = Affine<Fp>;
using G2Affine = Affine<Fp2>;
using G1XYZZ = XYZZ<Fp>;
using G2XYZZ = XYZZ<Fp2>;

// k * p, k = 8 x u32 canonical little-endian scalar (double-and-add, MSB first)
template <class F>
ZK_HD XYZZ<F> scalar_mul(const Affine<F>& p, const uint32_t k[8]) {
    XYZZ<F> r = XYZZ<F>::inf();
    for (int i = 255; i >= 0; i--) {
        r.dbl();
        if ((k[i >> 5] >> (i & 31)) & 1) r.madd(p);
    }
    return r;
}

// sum_t k_t P_t for a handful of terms (cnt <= 8): interleaved 4-bit windows -- ONE chain of 252 doublings shared by all terms, 15 precomputed multiples per
// point -- instead of one double-and-add per term.  Host use (digests the PLONK prover derives from digests); k_t canonical little-endian, below 2^256.
template <class F>
ZK_HD XYZZ<F> multi_scalar_mul(const Affine<F>* pts, const uint32_t (*k)[8], int cnt) {
    constexpr int MAXT = 8;
    XYZZ<F> tab[MAXT][16];
    if (cnt > MAXT) cnt = MAXT;
    for (int t = 0; t < cnt; t++) {
        tab[t][0] = XYZZ<F>::inf();
        tab[t][1] = pts[t].is_inf() ? XYZZ<F>::inf() : XYZZ<F>::from_affine(pts[t]);
        for (int j = 2; j < 16; j++) {
            tab[t][j] = tab[t][j - 1];
            tab[t][j].madd(pts[t]);  // (an affine (0, 0) -- infinity -- is skipped by madd)
        }
    }
    XYZZ<F> r = XYZZ<F>::inf();
    for (int w = 63; w >= 0; w--) {
        if (w != 63)
            for (int d = 0; d < 4; d++) r.dbl();
        for (int t = 0; t < cnt; t++) {
            const uint32_t dg = (k[t][w >> 3] >> (4 * (w & 7))) & 15;
            if (dg) r.add(tab[t][dg]);
        }
    }
    return r;
}

}  // namespace zkmi
