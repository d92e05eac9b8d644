// Host-only self test of the arithmetic that is shared between host and device builds (runs WITHOUT a GPU):
// the 32-bit-limb product-scanning Montgomery schedule of ff.hpp (host path of the same algorithm the gfx950 asm
// implements) and the XYZZ formulas of curve.hpp are checked against the independent 64-bit-limb HField code.
#include <string.h>

#include "acir_host.hpp"
#include "ctx.hpp"
#include "curve.hpp"
#include "host_ff.hpp"
#include "keyio.hpp"
#include "proofio.hpp"

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
    bad += !siphash_selftest();  // the content key's PRF against the paper's and the reference implementation's vectors (acir_host.hpp)
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
    // interleaved-window multi-scalar multiplication (curve.hpp; the PLONK prover's host-side digests) against the sum of plain double-and-add products: random
    // scalars, a zero scalar, a point at infinity, the same point twice with opposite scalars' worth (k and r - k would need Fr here: equal points suffice)
    {
        Affine<HFp> pts[5] = {p1.to_affine(), p2.to_affine(), p3.to_affine(), Affine<HFp>::inf(), p1.to_affine()};
        uint32_t ks[5][8];
        for (int t = 0; t < 5; t++) {
            HFr r = rnd<HFrParams>(s);
            for (int i = 0; i < 4; i++) { ks[t][2 * i] = (uint32_t)r.l[i]; ks[t][2 * i + 1] = (uint32_t)(r.l[i] >> 32); }
        }
        for (int i = 0; i < 8; i++) ks[1][i] = 0;
        ks[2][0] = 1;
        for (int i = 1; i < 8; i++) ks[2][i] = 0;
        XYZZ<HFp> want = XYZZ<HFp>::inf();
        for (int t = 0; t < 5; t++) want.add(scalar_mul(pts[t], ks[t]));
        bad += !same(multi_scalar_mul(pts, ks, 5).to_affine(), want.to_affine());
        bad += !multi_scalar_mul(pts, ks, 0).is_inf();
        bad += !same(multi_scalar_mul(pts + 2, ks + 2, 1).to_affine(), pts[2]);
    }
    // the transcript primitives of the PLONK prover (proofio.hpp): SHA-256 known answers (FIPS 180-4: "abc", the 56-byte message that needs two
    // blocks, one million 'a'), the transcript's chaining, fr.SetBytes' reduction
    {
        auto hex = [](const uint8_t d[32]) { static const char* x = "0123456789abcdef"; std::string o; for (int i = 0; i < 32; i++) { o += x[d[i] >> 4]; o += x[d[i] & 15]; } return o; };
        uint8_t dg[32];
        Sha256 h;
        h.update("abc", 3);
        h.final(dg);
        bad += hex(dg) != "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad";
        h.reset();
        h.update("abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq", 56);
        h.final(dg);
        bad += hex(dg) != "248d6a61d20638b8e5c026930c3e6039a33ce45964ff2167f6ecedd419db06c1";
        h.reset();
        std::string a1000(1000, 'a');
        for (int i = 0; i < 1000; i++) h.update(a1000.data(), 1000);
        h.final(dg);
        bad += hex(dg) != "cdc76e5c9914fb9281a1c7e284d73e67f1809a48a497200e046d39ccc7112cd0";
        // challenge_1 = H("beta" || challenge_0), challenge_0 = H("gamma" || bindings)
        FsTranscript t{"gamma", "beta"};
        t.bind(0, "\x01\x02", 2);
        (void)t.challenge(0);
        Sha256 g0;
        g0.update("gamma\x01\x02", 7);
        g0.final(dg);
        bad += memcmp(dg, t.value[0].data(), 32) != 0;
        (void)t.challenge(1);
        Sha256 g1;
        g1.update("beta", 4);
        g1.update(dg, 32);
        uint8_t d1[32];
        g1.final(d1);
        bad += memcmp(d1, t.value[1].data(), 32) != 0;
        // SetBytes reduces: 2^256 - 1 mod r == (2^256 mod r) - 1, and a canonical value survives a round trip
        uint8_t ff[32];
        memset(ff, 0xff, 32);
        HFr top = fr_from_be_reduce(ff);
        bad += !((top + HFr::one()) == HFr::one().to_mont());  // (x * R) with x = 2^256 mod r is to_mont(R mod r) ... i.e. one() * R
        uint8_t be[32];
        HFr v = rnd<HFrParams>(s).to_mont();
        fr_to_be(v, be);
        bad += !(fr_from_be_reduce(be) == v);
    }
    // G2 decompression on the host (keyio.hip): the compressed generator round-trips and a flipped x is rejected
    {
        Affine<HFp2> gen2;
        static const uint64_t X0[4] = {0x46debd5cd992f6edULL, 0x674322d4f75edaddULL, 0x426a00665e5c4479ULL, 0x1800deef121f1e76ULL};
        static const uint64_t X1[4] = {0x97e485b7aef312c2ULL, 0xf1aa493335a9e712ULL, 0x7260bfb731fb5d25ULL, 0x198e9393920d483aULL};
        static const uint64_t Y0[4] = {0x4ce6cc0166fa7daaULL, 0xe3d1e7690c43d37bULL, 0x4aab71808dcb408fULL, 0x12c85ea5db8c6debULL};
        static const uint64_t Y1[4] = {0x55acdadcd122975bULL, 0xbc4b313370b38ef3ULL, 0xec9e99ad690c3395ULL, 0x090689d0585ff075ULL};
        gen2.x.a0 = HFp{{X0[0], X0[1], X0[2], X0[3]}}.to_mont(); gen2.x.a1 = HFp{{X1[0], X1[1], X1[2], X1[3]}}.to_mont();
        gen2.y.a0 = HFp{{Y0[0], Y0[1], Y0[2], Y0[3]}}.to_mont(); gen2.y.a1 = HFp{{Y1[0], Y1[1], Y1[2], Y1[3]}}.to_mont();
        uint8_t enc[64];
        g2_compress(gen2, enc);
        Affine<HFp2> back;
        bad += !g2_decompress_host(enc, &back);
        bad += !same(back, gen2);
        enc[63] ^= 1;
        bad += g2_decompress_host(enc, &back) ? 1 : 0;  // another x: not on the twist, or outside the r-torsion
    }
    return bad;
}
