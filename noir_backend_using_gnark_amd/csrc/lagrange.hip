// The KZG SRS in LAGRANGE form over a key's domain: LagSRS[i] = [L_i(tau)]_1 = (1/n) sum_j w^(-ij) [tau^j]_1 -- the inverse DFT of the SRS's first n points,
// taken "in the exponent" (a radix-2 transform whose butterflies add points and multiply them by roots of unity).
//
// Why: plonk.Prove commits l, r, o as kzg.Commit(canonical coefficients) (gnark v0.8.0 backend/plonk/bn254/prove.go, reached from
// gnark_backend_ffi/backend/plonk/plonk.go:53-73).  The canonical coefficients of a wire polynomial are uniform field elements whatever the circuit computes,
// while its EVALUATIONS are the wire values themselves -- bits, bytes, 32-bit words in any real circuit.  Both give the same group element:
//     [l(tau)] = sum_j c_j [tau^j] = sum_i l_i [L_i(tau)],
// and the blinding (b0 + b1 X)(X^n - 1) adds b0 [tau^n - 1] + b1 [tau^(n+1) - tau].  With the two extra points appended, the three commitments of round 1 are ONE
// batched multi-exp (msm.hip) over wire values: a quarter to a third of the additions of the same multi-exp over coefficients (zeros vanish, ones meet in one
// bucket, small values touch two windows of thirteen), and it no longer waits for the three inverse transforms.  Same proof bytes.
//
// Cost, once per (SRS, domain): n/2 log2 n + n point-by-scalar multiplications (0.15 s at 2^19, 1.3 s at 2^22) -- the price of not knowing tau.  So this is for
// a prover that keeps a key (zk_bn254_plonk_pk_lagrange_srs); a process that makes one proof never builds it.
#include <stdint.h>

#include <vector>

#include "ctx.hpp"
#include "curve.hpp"
#include "ff29.hpp"
#include "host_ff.hpp"
#include "keyio.hpp"
#include "lagrange.hpp"
#include "multidev.hpp"
#include "msm.hpp"
#include "ntt.hpp"
#include "proofio.hpp"

namespace zkmi {

// k * P for a 254-bit k (canonical limbs): double-and-add from the top bit in the 29-bit-limb XYZZ arithmetic of the bucket reductions (ff29.hpp acc29_dbl /
// acc29_add: every special case of the group law handled).  Lanes of a wave that hold the same k take the same branches; the launches below arrange that
// wherever the transform allows it.
__device__ __forceinline__ XYZZ<Fp> g1_scalar_mul29(const XYZZ<Fp>& P, const Fr& k) {
    if (P.is_inf()) return XYZZ<Fp>::inf();
    Acc29 B, A;
    acc29_from_xyzz(B, P);
    A = B;
    A.inf = true;
    int top = 7;
    while (top >= 0 && k.l[top] == 0) top--;
    if (top < 0) return XYZZ<Fp>::inf();
    int bit = 31 - __clz(k.l[top]);
    for (int w = top; w >= 0; w--) {
        const uint32_t limb = k.l[w];
        for (int b = (w == top ? bit : 31); b >= 0; b--) {
            acc29_dbl(A);
            if ((limb >> b) & 1) acc29_add(A, B);
        }
    }
    return acc29_to_xyzz(A);
}
__device__ __forceinline__ XYZZ<Fp> g1_add29(const XYZZ<Fp>& a, const XYZZ<Fp>& b) {
    Acc29 A, B;
    acc29_from_xyzz(A, a);
    acc29_from_xyzz(B, b);
    acc29_add(A, B);
    return acc29_to_xyzz(A);
}
__device__ __forceinline__ XYZZ<Fp> g1_neg(const XYZZ<Fp>& a) {
    XYZZ<Fp> r = a;
    r.y = Fp::zero() - a.y;
    return r;
}
template <class T>
__device__ __forceinline__ T lg_load(const T* p) {
    static_assert(sizeof(T) % 16 == 0, "16-byte multiples only");
    T r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4* d = reinterpret_cast<uint4*>(&r);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 16; i++) d[i] = q[i];
    return r;
}
template <class T>
__device__ __forceinline__ void lg_store(T* p, const T& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    const uint4* d = reinterpret_cast<const uint4*>(&v);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 16; i++) q[i] = d[i];
}

// out[i] = s * P_i (affine in, XYZZ out): the 1/n of the inverse transform, applied to the inputs (one scalar for every lane)
__global__ __launch_bounds__(256) void k_lag_scale_in(const Affine<Fp>* __restrict__ pts, uint32_t n, Fr s, XYZZ<Fp>* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<Fp> p = lg_load(pts + i);
    lg_store(out + i, p.is_inf() ? XYZZ<Fp>::inf() : g1_scalar_mul29(XYZZ<Fp>::from_affine(p), s));
}
// One decimation-in-frequency stage, in place: blocks of 2m points, butterfly (i, i + m) with i = blk * 2m + k:
//     a' = a + b,   b' = (a - b) * w^(-k * n / (2m))        (tw_inv[e] = w^(-e), e < n/2, Montgomery)
// Lane t handles k = t / blocks, blk = t % blocks: consecutive lanes are consecutive BLOCKS of one k, so that a wave shares its twiddle (and the branches of the
// scalar multiplication) whenever the stage has at least 64 blocks; a point is 128 B, so the stride between lanes costs no bandwidth.
__global__ __launch_bounds__(256) void k_lag_dif_stage(XYZZ<Fp>* __restrict__ pts, uint32_t n, uint32_t m, const Fr* __restrict__ tw_inv) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n / 2) return;
    const uint32_t blocks = n / (2 * m), k = t / blocks, blk = t - k * blocks;
    const size_t i = (size_t)blk * 2 * m + k, j = i + m;
    const XYZZ<Fp> a = lg_load(pts + i), b = lg_load(pts + j);
    lg_store(pts + i, g1_add29(a, b));
    const XYZZ<Fp> d = g1_add29(a, g1_neg(b));
    const uint32_t e = k * blocks;  // k * n / (2m) < n / 2
    if (e == 0) { lg_store(pts + j, d); return; }
    const Fr w = lg_load(tw_inv + e).from_mont();
    lg_store(pts + j, g1_scalar_mul29(d, w));
}
// the transform leaves index bitrev(i) where i belongs; affine again (one inversion per point), plus the two points of the blinding at n and n + 1:
//     out[n] = [tau^n] - [1],  out[n + 1] = [tau^(n+1)] - [tau]      (srs holds at least n + 2 points)
__global__ __launch_bounds__(256) void k_lag_finish(const XYZZ<Fp>* __restrict__ pts, const Affine<Fp>* __restrict__ srs, uint32_t n, unsigned logn,
                                                    Affine<Fp>* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n + 2) return;
    if (i < n) {
        const uint32_t r = logn ? __brev(i) >> (32 - logn) : 0;
        lg_store(out + i, lg_load(pts + r).to_affine());
        return;
    }
    const Affine<Fp> hi = lg_load(srs + i), lo = lg_load(srs + (i - n));  // tau^n with 1, tau^(n+1) with tau
    const XYZZ<Fp> H = hi.is_inf() ? XYZZ<Fp>::inf() : XYZZ<Fp>::from_affine(hi);
    const XYZZ<Fp> L = lo.is_inf() ? XYZZ<Fp>::inf() : XYZZ<Fp>::from_affine(lo);
    lg_store(out + i, g1_add29(H, g1_neg(L)).to_affine());
}

// d_srs: the SRS's G1 points (affine, at least 2^logn + 2 of them) -> *handle: n + 2 registered bases (with their window table when the planner gives one)
int lagrange_srs_build(const void* d_srs, size_t srs_n, unsigned logn, uint64_t* handle) {
    const size_t n = (size_t)1 << logn;
    if (logn > 27 || srs_n < n + 2) return set_err(ZK_ERR_ARG, "the Lagrange form over 2^%u points needs %zu SRS points, %zu given", logn, n + 2, srs_n);
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    Slot* s = g.s;
    hipStream_t st = s->stream;
    Domain* d;
    ZK_TRY(get_domain(s, st, logn ? logn : 1, DOM_TW_INV, &d));
    XYZZ<Fp>* work = nullptr;
    Affine<Fp>* out = nullptr;
    ZK_HIP(hipMalloc(&work, n * sizeof(XYZZ<Fp>)));
    struct Free { void* p; ~Free() { if (p) (void)hipFree(p); } } f1{work}, f2{nullptr};
    ZK_HIP(hipMalloc(&out, (n + 2) * sizeof(Affine<Fp>)));
    f2.p = out;
    const HFr ninv = d->card_inv;  // 1 / 2^logn (for logn = 0 the domain above is the 2-point one: take 1)
    Fr s_in;
    {
        const HFr v = logn ? ninv : HFr::one();
        uint32_t c[8];
        to_canonical_u32(v, c);
        for (int i = 0; i < 8; i++) s_in.l[i] = c[i];
    }
    const unsigned grid_n = (unsigned)((n + 255) / 256);
    ZK_LAUNCH(s, st, "lagrange_scale_in", k_lag_scale_in, dim3(grid_n), dim3(256), 0, (const Affine<Fp>*)d_srs, (uint32_t)n, s_in, work);
    for (size_t m = n / 2; m >= 1; m >>= 1)
        ZK_LAUNCH(s, st, "lagrange_dif_stage", k_lag_dif_stage, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, work, (uint32_t)n, (uint32_t)m, (const Fr*)d->tw_inv);
    ZK_LAUNCH(s, st, "lagrange_finish", k_lag_finish, dim3((unsigned)((n + 2 + 255) / 256)), dim3(256), 0, (const XYZZ<Fp>*)work, (const Affine<Fp>*)d_srs, (uint32_t)n, logn, out);
    ZK_TRY(slot_sync(s, st));
    return zk_bn254_bases_register_dev(out, n + 2, 0, handle);
}

}  // namespace zkmi

// The Lagrange form of a registered G1 base array over the domain of 2^log_n points, as a base array of its own (2^log_n + 2 points: [L_i] for i < n, then
// P_n - P_0 and P_(n+1) - P_1 -- for an SRS [tau^n - 1], [tau^(n+1) - tau]): sum_i e_i out[i] = sum_j c_j in[j] whenever c = FFTInverse(e).  Any points do (the
// map is linear); the caller frees the result with zk_bn254_bases_free.
extern "C" int zk_bn254_bases_lagrange(uint64_t bases, uint32_t log_n, uint64_t* out_handle) {
    using namespace zkmi;
    if (!out_handle) return set_err(ZK_ERR_ARG, "null pointer");
    if (md_is_composite(bases)) return set_err(ZK_ERR_ARG, "the Lagrange form needs the base array on one device entry");
    ZK_ON_ENTRY_OF(bases);
    const void* d = nullptr;
    size_t n = 0;
    int is_g2 = 0;
    ZK_TRY(bases_ptr(bases, &d, &n, &is_g2));
    if (is_g2) return set_err(ZK_ERR_ARG, "the Lagrange form is built for G1 base arrays");
    return lagrange_srs_build(d, n, log_n, out_handle);
}
