// Host-only self test of the arithmetic that is shared between host and device builds (runs WITHOUT a GPU):
// the 32-bit-limb product-scanning Montgomery schedule of ff.hpp (host path of the same algorithm the gfx950 asm
// implements) and the XYZZ formulas of curve.hpp are checked against the independent 64-bit-limb HField code.
#include <string.h>

#include "ctx.hpp"
#include "curve.hpp"
#include "host_ff.hpp"

namespace zkmi {
static uint64_t sm_next(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
template <class P>
static HField<P> rnd(uint64_t& s) {
    HField<P> x;
    for (int i = 0; i < 4; i++) x.l[i] = sm_next(s);
    x.l[3] &= 0x0fffffffffffffffULL;  // < 2^252 < p
    return x;
}
template <class A, class B>
static bool same(const A& a, const B& b) {
    static_assert(sizeof(A) == sizeof(B), "size");
    return memcmp(&a, &b, sizeof(A)) == 0;
}
template <class A, class B>
static A cast(const B& b) {
    A a;
    memcpy(&a, &b, sizeof(A));
    return a;
}
}  // namespace zkmi

using namespace zkmi;

extern "C" int zk_selftest_host(void) {
    int bad = 0;
    uint64_t s = 0x5e1f7e57;
    for (int it = 0; it < 200; it++) {
        HFp a = rnd<HFpParams>(s), b = rnd<HFpParams>(s);
        Fp a32 = cast<Fp>(a), b32 = cast<Fp>(b);
        bad += !same(a * b, a32 * b32);
        bad += !same(a + b, a32 + b32);
        bad += !same(a - b, a32 - b32);
        bad += !same(b - a, b32 - a32);
        bad += !same(a.neg(), a32.neg());
        HFr c = rnd<HFrParams>(s), d = rnd<HFrParams>(s);
        Fr c32 = cast<Fr>(c), d32 = cast<Fr>(d);
        bad += !same(c * d, c32 * d32);
        bad += !same(c.to_mont().from_mont(), c32);
    }
    {
        HFp a = rnd<HFpParams>(s);
        bad += !same(a.inv(), cast<Fp>(a).inv());
        bad += !((a * a.inv()) == HFp::one());
    }
    // curve: k*G by both implementations, mixed/full additions, doubling
    Affine<HFp> g{HFp::one(), HFp::one() + HFp::one()};
    Affine<Fp> g32 = cast<Affine<Fp>>(g);
    uint32_t k1[8] = {0x12345678u, 0x9abcdef0u, 0x0fedcba9u, 0x87654321u, 0x11111111u, 0x22222222u, 0x33333333u, 0x01234567u};
    uint32_t k2[8] = {7, 0, 0, 0, 0, 0, 0, 0};
    XYZZ<HFp> p1 = scalar_mul(g, k1), p2 = scalar_mul(g, k2);
    XYZZ<Fp> q1 = scalar_mul(g32, k1), q2 = scalar_mul(g32, k2);
    bad += !same(p1, q1);
    bad += !same(p2, q2);
    XYZZ<HFp> p3 = p1;
    p3.add(p2);
    XYZZ<Fp> q3 = q1;
    q3.add(q2);
    bad += !same(p3, q3);
    bad += !same(p3.to_affine(), q3.to_affine());
    // on-curve: y^2 = x^3 + 3
    Affine<HFp> a3 = p3.to_affine();
    HFp three = HFp::one() + HFp::one() + HFp::one();
    bad += !(a3.y.sqr() == a3.x.sqr() * a3.x + three);
    // P + (-P) = inf ; P + P = 2P
    XYZZ<HFp> z = p1;
    z.add(p1.neg());
    bad += !z.is_inf();
    XYZZ<HFp> dd = p1, d2 = p1;
    dd.add(p1);
    d2.dbl();
    bad += !same(dd.to_affine(), d2.to_affine());
    return bad;
}
