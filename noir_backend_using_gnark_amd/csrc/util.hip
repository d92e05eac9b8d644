// Deterministic synthetic inputs generated ON the device: at 2^24..2^26 points the bases cannot be
// fabricated on host cores in reasonable time.  Streams are SplitMix64, identical to oracle/bn254_ref.py `rand_felts`
// and oracle/bn254_oracle.c `orc_rand_fr` / `orc_g1_gen_points`, so small cases can be cross-checked bit for bit.
#include <string.h>

#include "ctx.hpp"
#include "curve.hpp"
#include "host_ff.hpp"

namespace zkmi {

__device__ __forceinline__ uint64_t splitmix_at(uint64_t seed, uint64_t k) {  // k-th output (1-based) of the stream
    uint64_t z = seed + k * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__device__ __forceinline__ bool geq_r(const uint32_t x[8]) {
    for (int i = 7; i >= 0; i--)
        if (x[i] != FrParams::MOD[i]) return x[i] > FrParams::MOD[i];
    return true;
}

// canonical Fr element i of the stream: 4 limbs -> reduce mod r (2^256 < 6r: at most 5 subtractions)
__device__ __forceinline__ Fr rand_fr_canonical(uint64_t seed, uint64_t i, int witness_like) {
    const uint64_t per = witness_like ? 5 : 4;
    Fr x;
    for (int k = 0; k < 4; k++) {
        uint64_t v = splitmix_at(seed, per * i + k + 1);
        x.l[2 * k] = (uint32_t)v;
        x.l[2 * k + 1] = (uint32_t)(v >> 32);
    }
    if (witness_like) {
        uint64_t sel = splitmix_at(seed, per * i + 5) & 3;
        if (sel < 2) {
            x.l[0] &= 1;
            for (int k = 1; k < 8; k++) x.l[k] = 0;
        } else if (sel == 2) {
            for (int k = 1; k < 8; k++) x.l[k] = 0;
        }
    }
    while (geq_r(x.l)) {
        uint64_t bw = 0;
        for (int k = 0; k < 8; k++) {
            uint64_t d = (uint64_t)x.l[k] - FrParams::MOD[k] - bw;
            x.l[k] = (uint32_t)d;
            bw = d >> 63;
        }
    }
    return x;
}

__global__ void k_fr_random(Fr* out, size_t n, uint64_t seed, int mont, int witness_like) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr x = rand_fr_canonical(seed, i, witness_like);
    if (mont) x = x.to_mont();
    out[i] = x;
}

// P_i = k_i * G, k_i = i-th element of the uniform stream; affine output (one Fermat inversion per thread)
template <class F>
__global__ __launch_bounds__(256) void k_generate_points(Affine<F>* out, size_t n, uint64_t seed, Affine<F> gen) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr k = rand_fr_canonical(seed, i, 0);
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int b = 255; b >= 0; b--) {
        acc.dbl();
        if ((k.l[b >> 5] >> (b & 31)) & 1) acc.madd(gen.x, gen.y);
    }
    out[i] = acc.to_affine();
}

// kzg.NewSRS(size, alpha) [gnark-crypto ecc/bn254/fr/kzg; the reference builds its SRS with it at gnark_backend_ffi/backend/common.go:137
// and main.go:176]: G1[i] = alpha^i * G1.  One lane per point: alpha^i from the bits of i (alpha^(2^b) precomputed), then double-and-add.
struct PowBits {
    Fr pw[28];
};
__global__ __launch_bounds__(256) void k_kzg_srs_g1(Affine<Fp>* out, size_t n, PowBits basis, Affine<Fp> gen) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr k = Fr::one();
    for (unsigned b = 0; b < 28; b++)
        if ((i >> b) & 1) k = k * basis.pw[b];
    k = k.from_mont();
    XYZZ<Fp> acc = XYZZ<Fp>::inf();
    for (int b = 255; b >= 0; b--) {
        acc.dbl();
        if ((k.l[b >> 5] >> (b & 31)) & 1) acc.madd(gen.x, gen.y);
    }
    out[i] = acc.to_affine();
}

static Affine<Fp> g1_generator() {
    Affine<Fp> g;
    g.x = Fp::one();
    g.y = Fp::one() + Fp::one();
    return g;
}
// G2 generator, Montgomery limbs (cross-checked by tests against oracle/bn254_ref.py)
static Affine<Fp2> g2_generator() {
    static const uint32_t X0[8] = {0xd992f6edu, 0x46debd5cu, 0xf75edaddu, 0x674322d4u, 0x5e5c4479u, 0x426a0066u, 0x121f1e76u, 0x1800deefu};
    static const uint32_t X1[8] = {0xaef312c2u, 0x97e485b7u, 0x35a9e712u, 0xf1aa4933u, 0x31fb5d25u, 0x7260bfb7u, 0x920d483au, 0x198e9393u};
    static const uint32_t Y0[8] = {0x66fa7daau, 0x4ce6cc01u, 0x0c43d37bu, 0xe3d1e769u, 0x8dcb408fu, 0x4aab7180u, 0xdb8c6debu, 0x12c85ea5u};
    static const uint32_t Y1[8] = {0xd122975bu, 0x55acdadcu, 0x70b38ef3u, 0xbc4b3133u, 0x690c3395u, 0xec9e99adu, 0x585ff075u, 0x090689d0u};
    Affine<Fp2> g;
    Fp t;
    memcpy(t.l, X0, 32); g.x.a0 = t.to_mont();
    memcpy(t.l, X1, 32); g.x.a1 = t.to_mont();
    memcpy(t.l, Y0, 32); g.y.a0 = t.to_mont();
    memcpy(t.l, Y1, 32); g.y.a1 = t.to_mont();
    return g;
}

}  // namespace zkmi

using namespace zkmi;

extern "C" {

int zk_bn254_fr_random_dev(void* d_out, size_t n, uint64_t seed, int mont, int witness_like, void* stream) {
    if (!d_out) return set_err(ZK_ERR_ARG, "null pointer");
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = stream ? (hipStream_t)stream : g.s->stream;
    if (n) ZK_LAUNCH(g.s, st, "fr_random", k_fr_random, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (Fr*)d_out, n, seed, mont, witness_like);
    if (!stream || ctx().profiling) ZK_TRY(slot_sync(g.s, st));
    return ZK_OK;
}

int zk_bn254_g1_generate_dev(void* d_out, size_t n, uint64_t seed, void* stream) {
    if (!d_out) return set_err(ZK_ERR_ARG, "null pointer");
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = stream ? (hipStream_t)stream : g.s->stream;
    if (n)
        ZK_LAUNCH(g.s, st, "g1_generate", (k_generate_points<Fp>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (Affine<Fp>*)d_out, n, seed,
                  g1_generator());
    if (!stream || ctx().profiling) ZK_TRY(slot_sync(g.s, st));
    return ZK_OK;
}

int zk_bn254_g2_generate_dev(void* d_out, size_t n, uint64_t seed, void* stream) {
    if (!d_out) return set_err(ZK_ERR_ARG, "null pointer");
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = stream ? (hipStream_t)stream : g.s->stream;
    if (n)
        ZK_LAUNCH(g.s, st, "g2_generate", (k_generate_points<Fp2>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (Affine<Fp2>*)d_out, n, seed,
                  g2_generator());
    if (!stream || ctx().profiling) ZK_TRY(slot_sync(g.s, st));
    return ZK_OK;
}

// kzg.NewSRS(size, alpha): size G1 points alpha^i * G1 into d_g1_out (device), and [G2, alpha * G2] to the host.
int zk_bn254_kzg_new_srs_dev(void* d_g1_out, size_t size, const zk_fr* alpha, zk_g2_affine g2_out[2], void* stream) {
    if ((size && !d_g1_out) || !alpha) return set_err(ZK_ERR_ARG, "null pointer");
    if (size > ((size_t)1 << 28)) return set_err(ZK_ERR_ARG, "SRS size %zu exceeds 2^28", size);
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = stream ? (hipStream_t)stream : g.s->stream;
    HFr a;
    memcpy(&a, alpha, 32);
    PowBits pb;
    for (int b = 0; b < 28; b++) {
        memcpy(&pb.pw[b], &a, 32);
        a = a.sqr();
    }
    if (size) ZK_LAUNCH(g.s, st, "kzg_srs_g1", k_kzg_srs_g1, dim3((unsigned)((size + 255) / 256)), dim3(256), 0, (Affine<Fp>*)d_g1_out, size, pb, g1_generator());
    if (g2_out) {
        Affine<Fp2> gd = g2_generator();
        Affine<HFp2> gh;
        memcpy(&gh, &gd, sizeof gh);
        HFr a0;
        memcpy(&a0, alpha, 32);
        HFr can = a0.from_mont();
        uint32_t k[8];
        memcpy(k, can.l, 32);
        Affine<HFp2> ag = scalar_mul(gh, k).to_affine();
        memcpy(&g2_out[0], &gh, 128);
        memcpy(&g2_out[1], &ag, 128);
    }
    if (!stream || ctx().profiling) ZK_TRY(slot_sync(g.s, st));
    return ZK_OK;
}

}  // extern "C"
