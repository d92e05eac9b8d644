// groth16.Verify and plonk.Verify on the HOST (no device work; SURVEY 8f keeps verification last: it is O(1) next to a proof -- a few G1 scalar
// multiplications, two to four Miller loops, one final exponentiation, ~10 ms on one core).  They exist so that the reference's exports have a
// complete counterpart: PlonkVerifyWithVK (gnark_backend_ffi/main.go:44-56 -> backend/plonk/plonk.go:28-51: DeserializeProof / DeserializeFelts /
// DeserializeVerifyingKey, InitKZG(srs), plonk.Verify) and the intended Groth16 VerifyWithVK (backend/groth16/r1cs.go:176-212).  Inputs are gnark's
// wire images: Proof.WriteTo, VerifyingKey.WriteTo ([UPSTREAM-RECALL], as restated in oracle/plonk_ref.py), public inputs as Montgomery fr.Elements.
// The equations are those of oracle/plonk_ref.plonk_verify / oracle/bn254_ref.groth16_verify, written independently on the host field types.
#include <string.h>

#include <vector>

#include "ctx.hpp"
#include "curve.hpp"
#include "host_ff.hpp"
#include "keyio.hpp"
#include "pairing.hpp"
#include "proofio.hpp"

namespace zkmi {
namespace {

typedef Affine<HFp> G1;
typedef Affine<HFp2> G2;

// G1Affine.SetBytes on a compressed encoding: flags, x < q, y = (x^3 + 3)^((q + 1) / 4), sign by the "largest" flag
bool g1_decompress_host(const uint8_t in[32], G1* out) {
    const unsigned flag = in[0] >> 6;
    *out = G1::inf();
    if (flag == 1) {
        if (in[0] & 0x3f) return false;
        for (int i = 1; i < 32; i++)
            if (in[i]) return false;
        return true;
    }
    if (flag == 0) return false;
    uint64_t t[4];
    for (int i = 0; i < 4; i++) {
        uint64_t v = 0;
        for (int b = 0; b < 8; b++) v = (v << 8) | (uint8_t)(in[8 * (3 - i) + b] & ((i == 3 && b == 0) ? 0x3f : 0xff));
        t[i] = v;
    }
    if (HFp::geq_mod(t)) return false;
    const HFp x = HFp{{t[0], t[1], t[2], t[3]}}.to_mont();
    const HFp three = HFp::one() + HFp::one() + HFp::one();
    const HFp rhs = x.sqr() * x + three;
    static const uint64_t E[4] = {0x4f082305b61f3f52ULL, 0x65e05aa45a1c72a3ULL, 0x6e14116da0605617ULL, 0x0c19139cb84c680aULL};  // (q + 1) / 4
    HFp y = rhs.pow(E);
    if (y.sqr() != rhs) return false;
    if (fp_lex_largest(y) != (flag == 3)) y = y.neg();
    out->x = x;
    out->y = y;
    return true;
}

XYZZ<HFp> g1_mul(const G1& p, const HFr& k) {
    uint32_t c[8];
    to_canonical_u32(k, c);
    return scalar_mul(p, c);
}
G1 g1_generator() { return G1{HFp::one(), HFp::one() + HFp::one()}; }
HFr fr_pow(HFr a, uint64_t e) {
    HFr r = HFr::one();
    while (e) {
        if (e & 1) r = r * a;
        a = a.sqr();
        e >>= 1;
    }
    return r;
}
uint64_t be64(const uint8_t* p) {
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v = (v << 8) | p[i];
    return v;
}
// bytes, or their hex text, into a vector
bool blob_bytes(const void* data, size_t len, int is_hex, std::vector<uint8_t>* out) {
    const uint8_t* p = (const uint8_t*)data;
    if (!is_hex) { out->assign(p, p + len); return true; }
    if (len & 1) return false;
    out->resize(len / 2);
    for (size_t i = 0; i < len / 2; i++) {
        int v = 0;
        for (int k = 0; k < 2; k++) {
            const int c = p[2 * i + k], d = (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : (c >= 'A' && c <= 'F') ? c - 'A' + 10 : -1;
            if (d < 0) return false;
            v = (v << 4) | d;
        }
        (*out)[i] = (uint8_t)v;
    }
    return true;
}
// e(C - v G1 + z H, [1]2) e(-H, [alpha]2) == 1   (kzg.Verify)
bool kzg_check(const G1& digest, const G1& h, const HFr& value, const HFr& point, const G2 g2[2]) {
    XYZZ<HFp> lhs = XYZZ<HFp>::from_affine(digest);
    lhs.add(g1_mul(g1_generator(), value).neg());
    lhs.add(g1_mul(h, point));
    return pairing::product_is_one({{lhs.to_affine(), g2[0]}, {h.neg(), g2[1]}});
}

}  // namespace
}  // namespace zkmi

using namespace zkmi;

extern "C" {

// groth16.Verify(proof, vk, publicWitness): e(Ar, Bs) == e(alpha, beta) e(sum_i w_i K_i, gamma) e(Krs, delta), w_0 = 1.
// proof: Proof.WriteTo (128 B); vk: VerifyingKey.WriteTo bytes or hex text ([alpha]1 [beta]1 [beta]2 [gamma]2 [delta]1 [delta]2 u32 len(K) K);
// public_inputs: n_public Montgomery fr.Elements WITHOUT the constant wire (gnark's public witness).  *accepted = 1 / 0; malformed encodings (what
// gnark's ReadFrom rejects) are ZK_ERR_ARG / ZK_ERR_LEN.
int zk_bn254_groth16_verify(const uint8_t proof[128], const void* vk, size_t vk_len, int vk_is_hex, const zk_fr* public_inputs, size_t n_public, int* accepted) {
    if (!proof || !vk || !accepted || (n_public && !public_inputs)) return set_err(ZK_ERR_ARG, "null pointer");
    *accepted = 0;
    std::vector<uint8_t> kb;
    if (!blob_bytes(vk, vk_len, vk_is_hex, &kb)) return set_err(ZK_ERR_ARG, "verifying key: invalid hex text");
    if (kb.size() < 292) return set_err(ZK_ERR_LEN, "verifying key: %zu bytes, at least 292 expected", kb.size());
    const size_t nk = ((size_t)kb[288] << 24) | ((size_t)kb[289] << 16) | ((size_t)kb[290] << 8) | kb[291];
    if (kb.size() != 292 + 32 * nk) return set_err(ZK_ERR_LEN, "verifying key: %zu bytes, the count says %zu K points (%zu bytes)", kb.size(), nk, 292 + 32 * nk);
    G1 alpha, beta1, delta1, ar, krs;
    G2 beta, gamma, delta, bs;
    if (!g1_decompress_host(kb.data(), &alpha) || !g1_decompress_host(kb.data() + 32, &beta1) || !g1_decompress_host(kb.data() + 192, &delta1))
        return set_err(ZK_ERR_ARG, "verifying key: invalid G1 point");
    if (!g2_decompress_host(kb.data() + 64, &beta) || !g2_decompress_host(kb.data() + 128, &gamma) || !g2_decompress_host(kb.data() + 224, &delta))
        return set_err(ZK_ERR_ARG, "verifying key: invalid G2 point");
    std::vector<G1> K(nk);
    for (size_t i = 0; i < nk; i++)
        if (!g1_decompress_host(kb.data() + 292 + 32 * i, &K[i])) return set_err(ZK_ERR_ARG, "verifying key: invalid K point %zu", i);
    if (!g1_decompress_host(proof, &ar) || !g2_decompress_host(proof + 32, &bs) || !g1_decompress_host(proof + 96, &krs))
        return set_err(ZK_ERR_ARG, "proof: invalid point encoding");
    if (nk != n_public + 1) return set_err(ZK_ERR_LEN, "invalid witness size, got %zu, expected %zu (public - ONE_WIRE)", n_public, nk ? nk - 1 : 0);  // upstream's message
    XYZZ<HFp> ic = XYZZ<HFp>::from_affine(K[0]);
    for (size_t i = 0; i < n_public; i++) {
        HFr w;
        memcpy(&w, &public_inputs[i], 32);
        ic.add(g1_mul(K[i + 1], w));
    }
    *accepted = pairing::product_is_one({{ar.neg(), bs}, {alpha, beta}, {ic.to_affine(), gamma}, {krs, delta}}) ? 1 : 0;
    return ZK_OK;
}

// plonk.Verify(proof, vk, publicWitness) of gnark v0.8.0: the four challenges re-derived from the transcript, the quotient identity at zeta, the
// linearised digest rebuilt from the verifying key, the batched opening folded with kzg's gamma, two KZG checks.
// proof: Proof.WriteTo (548 B); vk: plonk.VerifyingKey.WriteTo bytes or hex (368 B: Size, SizeInv, Generator, NbPublicVariables, CosetShift, S1..S3, Ql, Qr, Qm,
// Qo, Qk); srs_g2: the SRS's two G2 points ([1]2, [alpha]2 -- what InitKZG attaches); public_inputs: Montgomery.
int zk_bn254_plonk_verify(const uint8_t proof[548], const void* vk, size_t vk_len, int vk_is_hex, const zk_g2_affine srs_g2[2], const zk_fr* public_inputs, size_t n_public,
                          int* accepted) {
    if (!proof || !vk || !srs_g2 || !accepted || (n_public && !public_inputs)) return set_err(ZK_ERR_ARG, "null pointer");
    *accepted = 0;
    std::vector<uint8_t> kb;
    if (!blob_bytes(vk, vk_len, vk_is_hex, &kb)) return set_err(ZK_ERR_ARG, "verifying key: invalid hex text");
    if (kb.size() != 368) return set_err(ZK_ERR_LEN, "verifying key: %zu bytes, 368 expected", kb.size());
    const uint64_t n = be64(kb.data()), npub = be64(kb.data() + 72);
    if (n == 0 || (n & (n - 1)) || n > ((uint64_t)1 << 28)) return set_err(ZK_ERR_ARG, "verifying key: size %llu is not a power of two <= 2^28", (unsigned long long)n);
    const HFr size_inv = fr_from_be_reduce(kb.data() + 8), gen = fr_from_be_reduce(kb.data() + 40), u = fr_from_be_reduce(kb.data() + 80);
    G1 vkp[8];  // S1, S2, S3, Ql, Qr, Qm, Qo, Qk
    for (int k = 0; k < 8; k++)
        if (!g1_decompress_host(kb.data() + 112 + 32 * k, &vkp[k])) return set_err(ZK_ERR_ARG, "verifying key: invalid G1 point %d", k);
    G1 lro[3], z, h[3], batch_h, z_open_h;
    bool ok = true;
    for (int k = 0; k < 3; k++) ok = ok && g1_decompress_host(proof + 32 * k, &lro[k]) && g1_decompress_host(proof + 128 + 32 * k, &h[k]);
    ok = ok && g1_decompress_host(proof + 96, &z) && g1_decompress_host(proof + 224, &batch_h) && g1_decompress_host(proof + 484, &z_open_h);
    if (!ok) return set_err(ZK_ERR_ARG, "proof: invalid point encoding");
    if (proof[256] || proof[257] || proof[258] || proof[259] != 7) return set_err(ZK_ERR_ARG, "proof: the batched opening must carry 7 claimed values");
    HFr claimed[7];
    for (int k = 0; k < 7; k++) claimed[k] = fr_from_be_reduce(proof + 260 + 32 * k);
    const HFr zu = fr_from_be_reduce(proof + 516);
    if (npub != n_public) return set_err(ZK_ERR_LEN, "invalid witness size, got %zu, expected %llu", n_public, (unsigned long long)npub);
    std::vector<HFr> pub(n_public);
    if (n_public) memcpy(pub.data(), public_inputs, n_public * 32);
    G2 g2[2];
    memcpy(g2, srs_g2, sizeof g2);

    // challenges
    FsTranscript fs{"gamma", "beta", "alpha", "zeta"};
    for (int k = 0; k < 8; k++) fs.bind_g1(0, vkp[k]);
    for (const HFr& w : pub) fs.bind_fr(0, w);
    for (int k = 0; k < 3; k++) fs.bind_g1(0, lro[k]);
    const HFr gamma = fs.challenge(0), beta = fs.challenge(1);
    fs.bind_g1(2, z);
    const HFr alpha = fs.challenge(2);
    for (int k = 0; k < 3; k++) fs.bind_g1(3, h[k]);
    const HFr zeta = fs.challenge(3);

    // quotient identity at zeta
    const HFr one = HFr::one();
    const HFr zn = fr_pow(zeta, n), zz = zn - one;
    HFr pi = HFr::zero(), wi = one;
    for (size_t i = 0; i < n_public; i++) {  // PI(zeta) = sum_i L_i(zeta) w_i,  L_i(zeta) = w^i / n (zeta^n - 1) / (zeta - w^i)
        pi = pi + wi * size_inv * zz * (zeta - wi).inv() * pub[i];
        wi = wi * gen;
    }
    const HFr l1 = zz * size_inv * (zeta - one).inv();
    const HFr &quot = claimed[0], &lin_z = claimed[1], &lz = claimed[2], &rz = claimed[3], &oz = claimed[4], &s1z = claimed[5], &s2z = claimed[6];
    const HFr f1 = lz + beta * s1z + gamma, f2 = rz + beta * s2z + gamma;
    const HFr t = f1 * f2 * (oz + gamma) * alpha * zu;
    if (lin_z + pi + t - alpha * alpha * l1 != quot * zz) return ZK_OK;  // *accepted stays 0

    // folded quotient digest and linearised digest
    const HFr zp = fr_pow(zeta, n + 2);
    XYZZ<HFp> fh = g1_mul(h[2], zp);
    fh.madd(h[1]);
    fh = g1_mul(fh.to_affine(), zp);
    fh.madd(h[0]);
    const HFr uu = u * u;
    const HFr c_s3 = f1 * f2 * zu * beta * alpha;
    const HFr c_z = (lz + beta * zeta + gamma).neg() * (rz + beta * u * zeta + gamma) * (oz + beta * uu * zeta + gamma) * alpha + alpha * alpha * l1;
    XYZZ<HFp> lin = g1_mul(vkp[3], lz);           // Ql
    lin.add(g1_mul(vkp[4], rz));                  // Qr
    lin.add(g1_mul(vkp[5], lz * rz));             // Qm
    lin.add(g1_mul(vkp[6], oz));                  // Qo
    lin.madd(vkp[7]);                             // Qk
    lin.add(g1_mul(vkp[2], c_s3));                // S3
    lin.add(g1_mul(z, c_z));
    const G1 digests[7] = {fh.to_affine(), lin.to_affine(), lro[0], lro[1], lro[2], vkp[0], vkp[1]};
    // kzg.deriveGamma: one-challenge transcript over the point, the digests and the claimed values
    FsTranscript kt{"gamma"};
    kt.bind_fr(0, zeta);
    for (const G1& d : digests) kt.bind_g1(0, d);
    for (const HFr& v : claimed) kt.bind_fr(0, v);
    const HFr kg = kt.challenge(0);
    XYZZ<HFp> fd = XYZZ<HFp>::inf();
    HFr fe = HFr::zero(), acc = one;
    for (int k = 0; k < 7; k++) {
        fd.add(g1_mul(digests[k], acc));
        fe = fe + claimed[k] * acc;
        acc = acc * kg;
    }
    if (!kzg_check(fd.to_affine(), batch_h, fe, zeta, g2)) return ZK_OK;
    if (!kzg_check(z, z_open_h, zu, zeta * gen, g2)) return ZK_OK;
    *accepted = 1;
    return ZK_OK;
}

// e(P1, Q1) e(P2, Q2) ... == 1 for n pairs of affine Montgomery points (a building block for callers and tests: bilinearity is how the host pairing is
// checked against the oracle's independent implementation)
int zk_bn254_pairing_check(const zk_g1_affine* p, const zk_g2_affine* q, size_t n, int* is_one) {
    if ((n && (!p || !q)) || !is_one) return set_err(ZK_ERR_ARG, "null pointer");
    std::vector<std::pair<G1, G2>> v(n);
    for (size_t i = 0; i < n; i++) {
        memcpy(&v[i].first, &p[i], 64);
        memcpy(&v[i].second, &q[i], 128);
    }
    *is_one = pairing::product_is_one(v) ? 1 : 0;
    return ZK_OK;
}

}  // extern "C"
