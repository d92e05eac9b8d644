// Deterministic synthetic inputs generated ON the device: at 2^24..2^26 points the bases cannot be
// fabricated on host cores in reasonable time.  Streams are SplitMix64, identical to oracle/bn254_ref.py `rand_felts`
// and oracle/bn254_oracle.c `orc_rand_fr` / `orc_g1_gen_points`, so small cases can be cross-checked bit for bit.
#include <string.h>

#include <mutex>
#include <vector>

#include "ctx.hpp"
#include "curve.hpp"
#include "fixedbase.hpp"
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

// ---------------------------------------------------------------------------------------------- fixed-base scalar multiplication
// [k_i] G for many scalars and ONE base (the generator): 8-bit windows over a table T[w][d] = d * 2^(8w) * G (32 x 255 affine points: 0.5 MB for
// G1, 1 MB for G2, built once on the host), i.e. at most 32 mixed additions per point instead of 255 doublings + ~128 additions; every lane takes
// PTS points and converts them to affine with ONE shared inversion (Montgomery's trick).  Used by kzg.NewSRS, groth16.Setup and the synthetic
// point generator.  Scalar sources: an array of Montgomery scalars, the SplitMix64 stream, or the powers alpha^i.
struct PowBits {
    Fr pw[28];
};
struct ScalarSrc {
    int kind;            // 0: scalars[i] (Montgomery)   1: SplitMix64 stream `seed`, element i (canonical)   2: alpha^i from `basis` (alpha^(2^b), Montgomery)
    const Fr* scalars;
    uint64_t seed;
    PowBits basis;
};
__device__ __forceinline__ Fr scalar_at(const ScalarSrc& S, size_t i) {  // canonical (non-Montgomery) scalar i
    if (S.kind == 1) return rand_fr_canonical(S.seed, i, 0);
    if (S.kind == 2) {
        Fr k = Fr::one();
        for (unsigned b = 0; b < 28; b++)
            if ((i >> b) & 1) k = k * S.basis.pw[b];
        return k.from_mont();
    }
    const uint4* q = reinterpret_cast<const uint4*>(S.scalars + i);
    uint4 a = q[0], b = q[1];
    Fr k;
    k.l[0] = a.x; k.l[1] = a.y; k.l[2] = a.z; k.l[3] = a.w;
    k.l[4] = b.x; k.l[5] = b.y; k.l[6] = b.z; k.l[7] = b.w;
    return k.from_mont();
}
template <class F, int PTS>
__global__ __launch_bounds__(256) void k_fixed_base(ScalarSrc S, size_t n, const Affine<F>* __restrict__ table, Affine<F>* __restrict__ out) {
    const size_t T = (size_t)gridDim.x * blockDim.x, g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    XYZZ<F> acc[PTS];
    F pre[PTS];
    F run = F::one();
#pragma unroll
    for (int p = 0; p < PTS; p++) {
        const size_t i = g + (size_t)p * T;  // strided over the lanes: coalesced stores
        acc[p] = XYZZ<F>::inf();
        if (i < n) {
            Fr k = scalar_at(S, i);
            for (int w = 0; w < 32; w++) {
                uint32_t d = (k.l[w >> 2] >> (8 * (w & 3))) & 255u;
                if (d) {
                    Affine<F> t = table[w * 255 + (d - 1)];
                    acc[p].madd(t.x, t.y);
                }
            }
        }
        pre[p] = run;
        if (!acc[p].is_inf()) run = run * (acc[p].zz * acc[p].zzz);
    }
    F inv = run.inv();  // one inversion for the lane's PTS points
#pragma unroll
    for (int p = PTS - 1; p >= 0; p--) {
        const size_t i = g + (size_t)p * T;
        Affine<F> a = Affine<F>::inf();
        if (!acc[p].is_inf()) {
            F zi = inv * pre[p];                      // 1 / (zz * zzz) of point p
            inv = inv * (acc[p].zz * acc[p].zzz);
            a.x = acc[p].x * (zi * acc[p].zzz);
            a.y = acc[p].y * (zi * acc[p].zz);
        }
        if (i < n) out[i] = a;
    }
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

// T[w * 255 + (d - 1)] = d * 2^(8w) * gen, built on the host (8,160 points: ~10 ms for G1) and kept on the device for the life of the process
template <class HF, class F>
static int fixed_base_table(const Affine<F>& gen_dev, Affine<F>** d_out) {
    Affine<HF> gen;
    memcpy(&gen, &gen_dev, sizeof gen);
    std::vector<XYZZ<HF>> pts(32 * 255);
    XYZZ<HF> base = XYZZ<HF>::from_affine(gen);
    for (int w = 0; w < 32; w++) {
        XYZZ<HF> acc = XYZZ<HF>::inf();
        for (int d = 1; d < 256; d++) {
            acc.add(base);
            pts[w * 255 + d - 1] = acc;
        }
        for (int i = 0; i < 8; i++) base.dbl();
    }
    // to affine with ONE inversion (Montgomery's trick over zz * zzz; no entry is the point at infinity: d * 2^(8w) < r)
    std::vector<Affine<HF>> aff(pts.size());
    std::vector<HF> pre(pts.size());
    HF run = HF::one();
    for (size_t i = 0; i < pts.size(); i++) {
        pre[i] = run;
        run = run * (pts[i].zz * pts[i].zzz);
    }
    HF inv = run.inv();
    for (size_t i = pts.size(); i-- > 0;) {
        HF zi = inv * pre[i];
        inv = inv * (pts[i].zz * pts[i].zzz);
        aff[i] = Affine<HF>{pts[i].x * (zi * pts[i].zzz), pts[i].y * (zi * pts[i].zz)};
    }
    ZK_HIP(hipMalloc((void**)d_out, aff.size() * sizeof(Affine<F>)));
    ZK_HIP(hipMemcpy(*d_out, aff.data(), aff.size() * sizeof(Affine<F>), hipMemcpyHostToDevice));
    return ZK_OK;
}
static std::mutex g_fb_mu;
static Affine<Fp>* g_fb_g1s[MAX_ENTRIES] = {};   // per device entry
static Affine<Fp2>* g_fb_g2s[MAX_ENTRIES] = {};

// out[i] = [k_i] G1 / G2 for the scalar source S (declared in fixedbase.hpp for the other translation units)
int fixed_base_mul(Slot* s, hipStream_t st, int is_g2, const ScalarSrc& S, size_t n, void* d_out) {
    if (!n) return ZK_OK;
    Affine<Fp>*& g_fb_g1 = g_fb_g1s[current_entry()];
    Affine<Fp2>*& g_fb_g2 = g_fb_g2s[current_entry()];
    {
        std::lock_guard<std::mutex> lk(g_fb_mu);
        if (!is_g2 && !g_fb_g1) ZK_TRY((fixed_base_table<HFp, Fp>(g1_generator(), &g_fb_g1)));
        if (is_g2 && !g_fb_g2) ZK_TRY((fixed_base_table<HFp2, Fp2>(g2_generator(), &g_fb_g2)));
    }
    if (is_g2) {
        constexpr int PTS = 2;
        size_t lanes = (n + PTS - 1) / PTS;
        ZK_LAUNCH(s, st, "fixed_base_g2", (k_fixed_base<Fp2, PTS>), dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, S, n, (const Affine<Fp2>*)g_fb_g2, (Affine<Fp2>*)d_out);
    } else {
        constexpr int PTS = 4;
        size_t lanes = (n + PTS - 1) / PTS;
        ZK_LAUNCH(s, st, "fixed_base_g1", (k_fixed_base<Fp, PTS>), dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, S, n, (const Affine<Fp>*)g_fb_g1, (Affine<Fp>*)d_out);
    }
    return ZK_OK;
}
int fixed_base_mul_scalars(Slot* s, hipStream_t st, int is_g2, const Fr* d_scalars, size_t n, void* d_out) {
    ScalarSrc S = {};
    S.kind = 0;
    S.scalars = d_scalars;
    return fixed_base_mul(s, st, is_g2, S, n, d_out);
}
Affine<Fp> generator_g1() { return g1_generator(); }
Affine<Fp2> generator_g2() { return g2_generator(); }

}  // namespace zkmi

using namespace zkmi;

extern "C" {

int zk_bn254_fr_random_dev(void* d_out, size_t n, uint64_t seed, int mont, int witness_like, void* stream) {
    if (!d_out) return set_err(ZK_ERR_ARG, "null pointer");
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = stream ? (hipStream_t)stream : g.s->stream;
    if (n) ZK_LAUNCH(g.s, st, "fr_random", k_fr_random, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (Fr*)d_out, n, seed, mont, witness_like);
    if (!stream || profiling_on()) ZK_TRY(slot_sync(g.s, st));
    return ZK_OK;
}

int zk_bn254_g1_generate_dev(void* d_out, size_t n, uint64_t seed, void* stream) {
    if (!d_out) return set_err(ZK_ERR_ARG, "null pointer");
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = stream ? (hipStream_t)stream : g.s->stream;
    ScalarSrc S = {};
    S.kind = 1;
    S.seed = seed;
    ZK_TRY(fixed_base_mul(g.s, st, 0, S, n, d_out));
    if (!stream || profiling_on()) ZK_TRY(slot_sync(g.s, st));
    return ZK_OK;
}

int zk_bn254_g2_generate_dev(void* d_out, size_t n, uint64_t seed, void* stream) {
    if (!d_out) return set_err(ZK_ERR_ARG, "null pointer");
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = stream ? (hipStream_t)stream : g.s->stream;
    ScalarSrc S = {};
    S.kind = 1;
    S.seed = seed;
    ZK_TRY(fixed_base_mul(g.s, st, 1, S, n, d_out));
    if (!stream || profiling_on()) ZK_TRY(slot_sync(g.s, st));
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
    ScalarSrc S = {};
    S.kind = 2;
    for (int b = 0; b < 28; b++) {
        memcpy(&S.basis.pw[b], &a, 32);
        a = a.sqr();
    }
    ZK_TRY(fixed_base_mul(g.s, st, 0, S, size, d_g1_out));
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
    if (!stream || profiling_on()) ZK_TRY(slot_sync(g.s, st));
    return ZK_OK;
}

}  // extern "C"
