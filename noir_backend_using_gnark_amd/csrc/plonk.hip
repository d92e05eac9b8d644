// PLONK setup and prove on gfx950: gnark v0.8.0 `plonk.Setup` / `plonk.Prove` (internal/backend/bn254/plonk/setup.go, prove.go; pinned at
// /root/reference/gnark_backend_ffi/go.mod:23) -- the reference's only LIVE prove path: PlonkProveWithPK (gnark_backend_ffi/main.go:24-37)
// -> plonk.Prove at backend/plonk/plonk.go:67, plonk.Setup at backend/plonk/plonk.go:21, over the gates the reference emits per ACIR
// arithmetic opcode (backend/plonk/sparse_r1cs.go:44-107: qL*xa + qR*xb + qO*xc + qM*xa*xb + qK = 0).
//
//   Prove, from the solver's output (the values of all variables) onwards  [UPSTREAM-RECALL for the step order]:
//     l, r, o        evaluateLROSmallDomain (gather), FFTInverse -> canonical, Blind(1), kzg.Commit x3           -> gamma, beta
//     z              iop.BuildRatioCopyConstraint: per-row products, batch inversion, prefix product; FFTInverse, Blind(2), Commit -> alpha
//     h              every polynomial on the coset of the big domain (4n): 5 coset FFTs per proof (the key's 9 are cached at setup),
//                    one fused pointwise kernel (gate + alpha * ordering + alpha^2 * (z-1) L1) / (X^n - 1), coset FFTInverse, Commit x3 -> zeta
//     openings       evaluations at zeta / omega*zeta (chunked Horner + block scans), kzg.Open of z (synthetic division as a suffix scan),
//                    linearised polynomial, folded quotient, kzg.BatchOpenSinglePoint (fold, divide, commit)
//   Everything above is device work -- NTTs (ntt.hip), MSMs over the resident SRS with its window tables (msm.hip) and the kernels
//   below -- except the SHA-256 Fiat-Shamir transcript and three G1 scalar multiplications, which stay on the host like upstream.
//   The prover's randomness (9 blinding scalars) is an input; the challenges are derived as upstream does or pinned by the caller.
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

#include "ctx.hpp"
#include "curve.hpp"
#include "ff29.hpp"
#include "host_ff.hpp"
#include "msm.hpp"
#include "keyio.hpp"
#include "lagrange.hpp"
#include "multidev.hpp"
#include "ntt.hpp"
#include "proofio.hpp"
#include "text_host.hpp"

namespace zkmi {

static Fr to_dev(const HFr& h) {
    Fr r;
    memcpy(&r, &h, 32);
    return r;
}
__device__ __forceinline__ Fr ld(const Fr* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
__device__ __forceinline__ unsigned brev(unsigned i, unsigned logn) { return logn ? (__brev(i) >> (32 - logn)) : 0; }

// ------------------------------------------------------------------------------------------------ small kernels
// evaluateLROSmallDomain: placeholders (l = public input, r = o = solution[0]), gates, padding (all solution[0])
__global__ void k_gather_lro(const Fr* __restrict__ sol, const uint32_t* __restrict__ xa, const uint32_t* __restrict__ xb, const uint32_t* __restrict__ xc,
                             uint32_t npub, uint32_t nc, uint32_t n, Fr* __restrict__ l, Fr* __restrict__ r, Fr* __restrict__ o) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t ia = 0, ib = 0, ic = 0;
    if (i < npub) ia = i;
    else if (i < npub + nc) { ia = xa[i - npub]; ib = xb[i - npub]; ic = xc[i - npub]; }
    l[i] = ld(sol + ia);
    r[i] = ld(sol + ib);
    o[i] = ld(sol + ic);
}
// Lagrange form of a selector on the small domain: [first x npub | coefficients of the gates | 0 ...]
__global__ void k_place_selector(Fr* __restrict__ dst, const Fr* __restrict__ src, uint32_t npub, uint32_t nc, uint32_t n, Fr first) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr v = Fr::zero();
    if (i < npub) v = first;
    else if (i < npub + nc) v = ld(src + (i - npub));
    dst[i] = v;
}
__global__ void k_fill(Fr* dst, size_t n, Fr v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = v;
}
// omega^k from the domain's half table (w^i, i < n/2): w^(k + n/2) = -w^k
__device__ __forceinline__ Fr omega_pow(const Fr* __restrict__ tw, uint32_t k, uint32_t n) {
    uint32_t h = n >> 1;
    if (k < h) return ld(tw + k);
    Fr v = ld(tw + (k - h));
    return Fr::zero() - v;
}
// S_j in Lagrange form: sigma[idx] = id(perm[idx]), id(p) = u^(p / n) * omega^(p mod n)  (setup.go ccomputePermutationPolynomials)
__global__ void k_sigma_lagrange(const uint32_t* __restrict__ perm, const Fr* __restrict__ tw, uint32_t n, Fr u, Fr uu, Fr* __restrict__ out) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 3 * n) return;
    uint32_t p = perm[idx], j = p / n, k = p - j * n;
    Fr w = omega_pow(tw, k, n);
    if (j == 1) w = w * u;
    else if (j == 2) w = w * uu;
    out[idx] = w;
}
// rows of the copy-constraint ratio (iop.BuildRatioCopyConstraint): num_i = prod_j (w_j(i) + beta * u^j * omega^i + gamma),
// den_i = prod_j (w_j(i) + beta * sigma_j(i) + gamma)
__global__ void k_z_terms(const Fr* __restrict__ l, const Fr* __restrict__ r, const Fr* __restrict__ o, const Fr* __restrict__ sig, const Fr* __restrict__ tw,
                          uint32_t n, Fr beta, Fr beta_u, Fr beta_uu, Fr gamma, Fr* __restrict__ num, Fr* __restrict__ den) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr lv = ld(l + i) + gamma, rv = ld(r + i) + gamma, ov = ld(o + i) + gamma;
    Fr w = omega_pow(tw, i, n);
    num[i] = (lv + beta * w) * (rv + beta_u * w) * (ov + beta_uu * w);
    den[i] = (lv + beta * ld(sig + i)) * (rv + beta * ld(sig + n + i)) * (ov + beta * ld(sig + 2 * (size_t)n + i));
}
// qk completed with the public inputs, on the big coset, WITHOUT transforming it per proof: completed qk = qk + sum_i d_i L_i with d_i = w_i - LQk[i] (the public
// rows), and L_i(x) = L_0(x w^-i): on the coset x_j = g W^j (w = W^rho, rho = 4 or 8) that is the key's L_0 table read rho i places earlier.  Tables are in bit-reversed layout:
// entry p belongs to j = bitrev(p).  Linear and exact, hence the same values as FFT(coset)(canonical(LQk with the public inputs)) -- for any input.
constexpr uint32_t PLONK_PI_DIRECT_MAX = 32;
__global__ __launch_bounds__(256) void k_qk_coset(const Fr* __restrict__ e_cqk, const Fr* __restrict__ e_l1, const Fr* __restrict__ sol, const Fr* __restrict__ lqk,
                                                  uint32_t npub, unsigned logN4, unsigned log_rho, Fr* __restrict__ out) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t N4 = (size_t)1 << logN4;
    if (p >= N4) return;
    const uint32_t j = __brev((uint32_t)p) >> (32 - logN4);
    Fr v = ld(e_cqk + p);
    for (uint32_t i = 0; i < npub; i++) {
        const Fr d = ld(sol + i) - ld(lqk + i);
        const uint32_t q = __brev((j - (i << log_rho)) & (uint32_t)(N4 - 1)) >> (32 - logN4);  // w = W^rho, rho = N4 / n
        v = v + d * ld(e_l1 + q);
    }
    out[p] = v;
}

// fr.BatchInvert: a[i] <- 1 / a[i] (0 stays 0); BINV_K elements per lane (strided, coalesced) share one Fermat inversion (~380 products, paid per
// WAVE whatever the lanes do): 3 + 380 / BINV_K products per element.  The prefix products of the forward sweep go through `scratch` (n elements, 160 B of
// traffic per element in all) instead of registers, which is what lets BINV_K be 32 (8 in registers: 1.9 ms at 2^22 elements; 32: see DESIGN 3.8).
constexpr int BINV_K = 32;
__global__ __launch_bounds__(256) void k_batch_inverse(Fr* __restrict__ a, size_t n, Fr* __restrict__ scratch) {
    const size_t T = (size_t)gridDim.x * blockDim.x, g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fr acc = Fr::one();
    for (int k = 0; k < BINV_K; k++) {
        const size_t i = g + k * T;
        if (i >= n) break;
        Fr v = ld(a + i);
        if (v.is_zero()) v = Fr::one();  // handled on the way back
        scratch[i] = acc;
        acc = acc * v;
    }
    Fr inv = acc.inv();
    for (int k = BINV_K - 1; k >= 0; k--) {
        const size_t i = g + k * T;
        if (i >= n) continue;
        const Fr v = ld(a + i);
        if (v.is_zero()) continue;  // a zero stays a zero and took no part in the product
        a[i] = inv * ld(scratch + i);
        inv = inv * v;
    }
}

// ------------------------------------------------------------------------------------------------ scans (K consecutive elements per lane)
// Exclusive prefix product z[0] = 1, z[i+1] = z[i] * num[i] * dinv[i]  (the rolling products + ratio of BuildRatioCopyConstraint).
// pass 1: lane totals, block-level inclusive scan (Hillis-Steele in LDS); pass 2: one block scans the block totals; pass 3: apply.
__global__ __launch_bounds__(256) void k_pscan_local(const Fr* __restrict__ num, const Fr* __restrict__ dinv, size_t n, uint32_t K, Fr* __restrict__ texcl,
                                                     Fr* __restrict__ btot) {
    __shared__ Fr sh[256];
    size_t gt = (size_t)blockIdx.x * 256 + threadIdx.x, base = gt * K;
    Fr tot = Fr::one();
    for (uint32_t k = 0; k < K; k++) {
        size_t i = base + k;
        if (i < n) tot = tot * (ld(num + i) * ld(dinv + i));
    }
    sh[threadIdx.x] = tot;
    __syncthreads();
    Fr inc = tot;
    for (unsigned d = 1; d < 256; d <<= 1) {
        Fr o = threadIdx.x >= d ? sh[threadIdx.x - d] : Fr::one();
        __syncthreads();
        inc = inc * o;
        sh[threadIdx.x] = inc;
        __syncthreads();
    }
    texcl[gt] = threadIdx.x ? sh[threadIdx.x - 1] : Fr::one();
    if (threadIdx.x == 255) btot[blockIdx.x] = inc;
}
__global__ __launch_bounds__(1024) void k_pscan_blocks(Fr* __restrict__ btot, uint32_t nb) {  // in place: btot[b] <- prod_{b' < b} btot[b']
    __shared__ Fr sh[1024];
    Fr carry = Fr::one();
    for (uint32_t c0 = 0; c0 < nb; c0 += 1024) {
        uint32_t b = c0 + threadIdx.x;
        Fr v = b < nb ? ld(btot + b) : Fr::one();
        sh[threadIdx.x] = v;
        __syncthreads();
        Fr inc = v;
        for (unsigned d = 1; d < 1024; d <<= 1) {
            Fr o = threadIdx.x >= d ? sh[threadIdx.x - d] : Fr::one();
            __syncthreads();
            inc = inc * o;
            sh[threadIdx.x] = inc;
            __syncthreads();
        }
        Fr excl = carry * (threadIdx.x ? sh[threadIdx.x - 1] : Fr::one());
        Fr total = sh[1023];
        __syncthreads();
        if (b < nb) btot[b] = excl;
        carry = carry * total;
    }
}
__global__ __launch_bounds__(256) void k_pscan_apply(const Fr* __restrict__ num, const Fr* __restrict__ dinv, size_t n, uint32_t K, const Fr* __restrict__ texcl,
                                                     const Fr* __restrict__ bexcl, Fr* __restrict__ z) {
    size_t gt = (size_t)blockIdx.x * 256 + threadIdx.x, base = gt * K;
    if (base >= n) return;
    Fr p = ld(bexcl + blockIdx.x) * ld(texcl + gt);
    for (uint32_t k = 0; k < K; k++) {
        size_t i = base + k;
        if (i >= n) break;
        z[i] = p;
        p = p * (ld(num + i) * ld(dinv + i));
    }
}

// Suffix recurrence S_i = f_i + a * S_(i+1) over a polynomial f of `len` coefficients: S_0 = f(a), and q_i = S_(i+1) are the coefficients of
// (f - f(a)) / (X - a) -- kzg.dividePolyByXminusA and polynomial evaluation in one structure.
// pass 1: lane Horner totals T_t, block-level suffix scan with multiplier A = a^K (A^d doubles per step); lane carry-in and block totals out.
__global__ __launch_bounds__(256) void k_hscan_local(const Fr* __restrict__ f, size_t len, uint32_t K, Fr a, Fr A, Fr* __restrict__ tcarry, Fr* __restrict__ btot) {
    __shared__ Fr sh[256];
    size_t gt = (size_t)blockIdx.x * 256 + threadIdx.x, base = gt * K;
    Fr acc = Fr::zero();
    for (int k = (int)K - 1; k >= 0; k--) {
        size_t i = base + (size_t)k;
        acc = acc * a;
        if (i < len) acc = acc + ld(f + i);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    Fr val = acc, Ad = A;
    for (unsigned d = 1; d < 256; d <<= 1) {
        Fr o = threadIdx.x + d < 256 ? sh[threadIdx.x + d] : Fr::zero();
        __syncthreads();
        val = val + Ad * o;
        sh[threadIdx.x] = val;
        __syncthreads();
        Ad = Ad.sqr();
    }
    if (tcarry) tcarry[gt] = threadIdx.x < 255 ? sh[threadIdx.x + 1] : Fr::zero();
    if (threadIdx.x == 0) btot[blockIdx.x] = val;
}
// Several polynomials at the SAME point in one launch (plonk.Prove evaluates l, r, o, s1, s2 and then foldedH, linPol at zeta: seven evaluations that are two
// launches each -- a lane-Horner pass and a single-workgroup block scan that is pure latency): blockIdx.y selects the polynomial.
constexpr int HSCAN_BATCH_MAX = 8;
struct HscanBatch {
    const Fr* f[HSCAN_BATCH_MAX];
    size_t len[HSCAN_BATCH_MAX];
};
__global__ __launch_bounds__(256) void k_hscan_local_batch(HscanBatch Bt, uint32_t K, Fr a, Fr A, uint32_t nb, Fr* __restrict__ btot) {
    __shared__ Fr sh[256];
    const Fr* __restrict__ f = Bt.f[blockIdx.y];
    const size_t len = Bt.len[blockIdx.y];
    size_t gt = (size_t)blockIdx.x * 256 + threadIdx.x, base = gt * K;
    Fr acc = Fr::zero();
    for (int k = (int)K - 1; k >= 0; k--) {
        size_t i = base + (size_t)k;
        acc = acc * a;
        if (i < len) acc = acc + ld(f + i);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    Fr val = acc, Ad = A;
    for (unsigned d = 1; d < 256; d <<= 1) {
        Fr o = threadIdx.x + d < 256 ? sh[threadIdx.x + d] : Fr::zero();
        __syncthreads();
        val = val + Ad * o;
        sh[threadIdx.x] = val;
        __syncthreads();
        Ad = Ad.sqr();
    }
    if (threadIdx.x == 0) btot[(size_t)blockIdx.y * nb + blockIdx.x] = val;
}
// pass 2: one block; btot[b] <- C_b = sum_{b' > b} btot[b'] * M^(b'-b-1) (M = A^256); *total = S_0 = f(a).
__global__ __launch_bounds__(1024) void k_hscan_blocks(Fr* __restrict__ btot, uint32_t nb, Fr M, Fr* __restrict__ total) {
    __shared__ Fr sh[1024];
    btot += (size_t)blockIdx.x * nb;  // batched form: one workgroup per polynomial, rows of nb block totals, results in total[blockIdx.x]
    if (total) total += blockIdx.x;
    Fr carry = Fr::zero();  // S at the start of the chunk above
    uint32_t nchunks = (nb + 1023) / 1024;
    Fr M1024 = M;
    for (int i = 0; i < 10; i++) M1024 = M1024.sqr();
    for (int c = (int)nchunks - 1; c >= 0; c--) {
        uint32_t b = (uint32_t)c * 1024 + threadIdx.x;
        Fr v = b < nb ? ld(btot + b) : Fr::zero();
        sh[threadIdx.x] = v;
        __syncthreads();
        Fr val = v, Md = M;
        for (unsigned d = 1; d < 1024; d <<= 1) {
            Fr o = threadIdx.x + d < 1024 ? sh[threadIdx.x + d] : Fr::zero();
            __syncthreads();
            val = val + Md * o;
            sh[threadIdx.x] = val;
            __syncthreads();
            Md = Md.sqr();
        }
        // val = sum_{u >= t in chunk} v_u M^(u-t); global S''_b = val + M^(1024 - t) * carry
        Fr next = threadIdx.x < 1023 ? sh[threadIdx.x + 1] : Fr::zero();   // S'' of the next block, chunk-local part
        uint32_t e = 1023 - threadIdx.x;                                     // M^e * carry completes it
        Fr pw = Fr::one(), bs = M;
        for (int i = 0; i < 10; i++) { if ((e >> i) & 1) pw = pw * bs; bs = bs.sqr(); }
        Fr C = next + pw * carry;
        Fr chunk_total = sh[0];
        __syncthreads();
        if (b < nb) btot[b] = C;
        carry = chunk_total + M1024 * carry;
    }
    if (threadIdx.x == 0 && total) *total = carry;
}
// pass 3: q_i = S_(i+1); q may alias f
__global__ __launch_bounds__(256) void k_hscan_apply(const Fr* f, size_t len, uint32_t K, Fr a, Fr A, const Fr* __restrict__ tcarry, const Fr* __restrict__ bC, Fr* q) {
    size_t gt = (size_t)blockIdx.x * 256 + threadIdx.x, base = gt * K;
    if (base >= len) return;
    uint32_t e = 255 - threadIdx.x;
    Fr pw = Fr::one(), bs = A;
    for (int i = 0; i < 8; i++) { if ((e >> i) & 1) pw = pw * bs; bs = bs.sqr(); }
    Fr S = ld(tcarry + gt) + pw * ld(bC + blockIdx.x);
    for (int k = (int)K - 1; k >= 0; k--) {
        size_t i = base + (size_t)k;
        if (i >= len) continue;
        Fr fv = ld(f + i);
        q[i] = S;
        S = fv + a * S;
    }
}

// ------------------------------------------------------------------------------------------------ quotient numerator on the big coset
struct QuotArgs {
    const Fr *el, *er, *eo, *ez, *eqk;                                 // the proof's polynomials (blinded l, r, o, z; completed qk)
    const Fr *ql, *qr, *qm, *qo, *s1, *s2, *s3, *l1, *id;              // the key's, all LagrangeCoset on the big domain, bit-reversed layout
    Fr* out;                                                           // may alias eqk
    Fr alpha, beta, gamma, beta_u, beta_uu;
    Fr xn_inv[8];                                                      // 1 / (x^n - 1): x^n takes rho <= 8 values on the coset
    unsigned logN4, log_rho;
};
// t(x) = [ gate(x) + alpha * (zs * prod(w + beta s_j + gamma) - z * prod(w + beta u^j x + gamma)) + alpha^2 (z - 1) L1 ] / (x^n - 1)
// (prove.go: fic / fo / fone / fm, then iop.DivideByXMinusOne); index i is the bit-reversed position of natural index j.
__global__ __launch_bounds__(256) void k_quotient(QuotArgs A) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t N4 = (size_t)1 << A.logN4;
    if (i >= N4) return;
    const unsigned j = brev((unsigned)i, A.logN4);
    const unsigned rho = 1u << A.log_rho;
    const unsigned js = (j + rho) & (unsigned)(N4 - 1);  // z(omega * x): omega = W^rho
    const size_t is = brev(js, A.logN4);
    Fr l = ld(A.el + i), r = ld(A.er + i), o = ld(A.eo + i), z = ld(A.ez + i), zs = ld(A.ez + is), x = ld(A.id + i);
    Fr gate = ld(A.ql + i) * l + ld(A.qr + i) * r + ld(A.qm + i) * (l * r) + ld(A.qo + i) * o + ld(A.eqk + i);
    Fr lg = l + A.gamma, rg = r + A.gamma, og = o + A.gamma;
    Fr a = (lg + A.beta * x) * (rg + A.beta_u * x) * (og + A.beta_uu * x) * z;
    Fr b = (lg + A.beta * ld(A.s1 + i)) * (rg + A.beta * ld(A.s2 + i)) * (og + A.beta * ld(A.s3 + i)) * zs;
    Fr one = (z - Fr::one()) * ld(A.l1 + i);
    Fr t = ((one * A.alpha + (b - a)) * A.alpha + gate) * A.xn_inv[j & (rho - 1)];
    A.out[i] = t;
}

// The same numerator with its 21 products in the 9 x 29-bit representation of ff29.hpp (Fr29: 206 instructions per product instead of ~305; the kernel is bound by
// them, not by its 8.6 GB).  Values loaded from memory are canonical images V = v 2^256; a product needs ONE operand in the multiplier form V << 5 (= v 2^261):
// u29r_mul(X, Y << 5) = X Y / 2^261 stays in the 2^256 domain.  So the second and third factor of each permutation product are BUILT in the 2^261 domain --
// (w << 5) + (gamma << 5) + u29r_mul(x << 5, beta << 5) -- and no in-register shift is ever needed.  Sums are plain limb additions; bounds (units of r; checked with
// tools/u29_ntt_model.py's bound classes): gate < 5.8, first factors < 3.2, 2^261-domain factors < 71.1 (re-normalised once: two operands with
// 31-bit limbs would overflow a 64-bit column), products < 2.4, the last sum < 8.1, t < 2.6 -> one partial reduction, canonical image out.
// The schedule below is restated statement by statement in tools/u29_ntt_model.py (quotient_schedule): bound propagation + exactness against the field formula,
// run by the CPU suite (tests/test_limb_models.py) -- edit both or the suite fails.
__global__ __launch_bounds__(256, 4) void k_quotient29(QuotArgs A) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t N4 = (size_t)1 << A.logN4;
    if (i >= N4) return;
    const unsigned j = brev((unsigned)i, A.logN4);
    const unsigned rho = 1u << A.log_rho;
    const unsigned js = (j + rho) & (unsigned)(N4 - 1);  // z(omega * x): omega = W^rho
    const size_t is = brev(js, A.logN4);
    const Fr fl = ld(A.el + i), fr = ld(A.er + i), fo = ld(A.eo + i), fz = ld(A.ez + i), fx = ld(A.id + i);
    const U29 l = u29_unpack(fl), r = u29_unpack(fr), o = u29_unpack(fo), z = u29_unpack(fz);
    // gate = ql l + qr r + qm (l r) + qo o + qk
    U29 gate = u29r_mul(l, u29r_load5(ld(A.ql + i)));
    gate = u29_add(gate, u29r_mul(r, u29r_load5(ld(A.qr + i))));
    gate = u29_add(gate, u29r_mul(u29r_mul(l, u29r_load5(fr)), u29r_load5(ld(A.qm + i))));
    gate = u29_add(gate, u29r_mul(o, u29r_load5(ld(A.qo + i))));
    gate = u29_add(gate, u29_unpack(ld(A.eqk + i)));
    const U29 g = u29_unpack(A.gamma), g5 = u29r_load5(A.gamma), x5 = u29r_load5(fx);
    // a = (l + g + beta x) (r + g + beta u x) (o + g + beta u^2 x) z
    U29 a = u29_add(u29_add(l, g), u29r_mul(u29_unpack(fx), u29r_load5(A.beta)));
    a = u29r_mul(a, u29_wnorm(u29_add(u29_add(u29r_load5(fr), g5), u29r_mul(x5, u29r_load5(A.beta_u)))));
    a = u29r_mul(a, u29_wnorm(u29_add(u29_add(u29r_load5(fo), g5), u29r_mul(x5, u29r_load5(A.beta_uu)))));
    a = u29r_mul(a, u29r_load5(fz));
    // b = (l + g + beta s1) (r + g + beta s2) (o + g + beta s3) z(omega x)
    const U29 b5 = u29r_load5(A.beta);
    U29 b = u29_add(u29_add(l, g), u29r_mul(u29_unpack(ld(A.s1 + i)), b5));
    b = u29r_mul(b, u29_wnorm(u29_add(u29_add(u29r_load5(fr), g5), u29r_mul(u29r_load5(ld(A.s2 + i)), b5))));
    b = u29r_mul(b, u29_wnorm(u29_add(u29_add(u29r_load5(fo), g5), u29r_mul(u29r_load5(ld(A.s3 + i)), b5))));
    b = u29r_mul(b, u29r_load5(ld(A.ez + is)));
    // t = ((one alpha + (b - a)) alpha + gate) / (x^n - 1),  one = (z - 1) L1
    const U29 a5 = u29r_load5(A.alpha);
    const U29 one = u29r_mul(u29r_sub<4>(z, u29_unpack(Fr::one())), u29r_load5(ld(A.l1 + i)));
    U29 t = u29_add(u29r_mul(one, a5), u29r_sub<4>(b, a));
    t = u29_add(u29r_mul(t, a5), gate);
    t = u29r_mul(t, u29r_load5(A.xn_inv[j & (rho - 1)]));
    A.out[i] = u29r_pack(u29r_reduce(t), true);
}

// linearised polynomial (prove.go computeLinearizedPolynomial), len = n + 3 coefficients
struct LinArgs {
    const Fr *bz, *s3, *ql, *qr, *qm, *qo, *cqk;
    Fr* out;
    Fr c_z, c_s3, alpha, rl, lz, rz, oz, lag;
    uint32_t n, len;
};
__global__ void k_linearized(LinArgs A) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.len) return;
    Fr z = ld(A.bz + i);
    Fr v = z * A.c_z;
    if (i < A.n) v = v + ld(A.s3 + i) * A.c_s3;
    v = v * A.alpha;
    if (i < A.n) v = v + ld(A.qm + i) * A.rl + ld(A.ql + i) * A.lz + ld(A.qr + i) * A.rz + ld(A.qo + i) * A.oz + ld(A.cqk + i);
    A.out[i] = v + z * A.lag;
}
// out = sum_k c_k * p_k (coefficient-wise; p_k has len_k coefficients)  -- folded quotient and kzg.BatchOpenSinglePoint's fold
struct FoldArgs {
    const Fr* p[8];
    uint32_t len[8];
    Fr c[8];
    int k;
    uint32_t n;
    Fr* out;
};
__global__ void k_fold(FoldArgs A) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    Fr v = Fr::zero();
    for (int k = 0; k < A.k; k++)
        if (i < A.len[k]) v = v + ld(A.p[k] + i) * A.c[k];
    A.out[i] = v;
}
// (*Polynomial).Blind: p[i] -= b_i, p[n + i] += b_i  (p[n + i] is zero before: the canonical form has n coefficients)
struct BlindArgs { Fr b[3]; int k; };
__global__ void k_blind(Fr* p, uint32_t n, BlindArgs B) {
    int i = threadIdx.x;
    if (i >= B.k) return;
    p[i] = ld(p + i) - B.b[i];
    p[n + i] = B.b[i];
}
// the blinding scalars behind a vector of n evaluations: p[n + i] = b_i -- the scalars of [tau^n - 1], [tau^(n+1) - tau] in a commitment against the Lagrange-form SRS
__global__ void k_blind_tail(Fr* p, uint32_t n, BlindArgs B) {
    int i = threadIdx.x;
    if (i < B.k) p[n + i] = B.b[i];
}
// *flag |= 1 if any of a[0 .. n) is non-zero
__global__ void k_any_nonzero(const Fr* __restrict__ a, size_t n, int* __restrict__ flag) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !ld(a + i).is_zero()) atomicOr(flag, 1);
}
// qk[i] = -(ql a + qr b + qo c + qm a b): synthetic satisfiable circuits (bench / tests)
__global__ void k_synth_qk(Fr* qk, const Fr* ql, const Fr* qr, const Fr* qo, const Fr* qm, const uint32_t* xa, const uint32_t* xb, const uint32_t* xc,
                           const Fr* sol, size_t nc) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    Fr a = ld(sol + xa[i]), b = ld(sol + xb[i]), c = ld(sol + xc[i]);
    qk[i] = Fr::zero() - (ld(ql + i) * a + ld(qr + i) * b + ld(qo + i) * c + ld(qm + i) * (a * b));
}

// pk.Permutation on the wire: 3n raw big-endian int64 (encoding/binary, no length prefix) <-> uint32 slots
__global__ void k_perm_from_be(const uint32_t* __restrict__ raw, size_t cnt, uint32_t* __restrict__ out, int* __restrict__ status) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    uint32_t hi = __builtin_bswap32(raw[2 * i]), lo = __builtin_bswap32(raw[2 * i + 1]);
    if (hi != 0 || lo >= cnt) { atomicOr(status, 8); lo = 0; }
    out[i] = lo;
}
__global__ void k_perm_to_be(const uint32_t* __restrict__ in, size_t cnt, uint32_t* __restrict__ raw) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    raw[2 * i] = 0;
    raw[2 * i + 1] = __builtin_bswap32(in[i]);
}

// ------------------------------------------------------------------------------------------------ resident key
struct PlonkPK {
    unsigned logn = 0, logN4 = 0, log_rho = 0;
    size_t n = 0, N4 = 0, n_public = 0, n_constraints = 0, n_vars = 0;
    uint64_t srs = 0;
    uint64_t lag_srs = 0;  // the SRS in Lagrange form over this key's domain + the two points of the blinding (lagrange.hip; zk_bn254_plonk_pk_lagrange_srs), or 0
    HFr gen, u, card_inv;
    // canonical (regular) polynomials of the key, n each, and LQk (Lagrange)
    Fr *ql = nullptr, *qr = nullptr, *qm = nullptr, *qo = nullptr, *cqk = nullptr, *lqk = nullptr, *s1 = nullptr, *s2 = nullptr, *s3 = nullptr;
    uint32_t* perm = nullptr;  // pk.Permutation (3n), kept for ProvingKey.WriteTo
    Fr* sig = nullptr;      // S1 | S2 | S3 in Lagrange form (3n): what BuildRatioCopyConstraint reads through pk.Permutation
    Fr* e_cqk = nullptr;    // CQk (qk WITHOUT the public inputs) as LagrangeCoset on the big domain: per proof only the public inputs' share is added (k_qk_coset)
    Fr* e[9] = {};          // ql, qr, qm, qo, s1, s2, s3, L1, id as LagrangeCoset on the big domain, bit-reversed layout (gnark caches the first 7)
    uint32_t *xa = nullptr, *xb = nullptr, *xc = nullptr;
    Affine<HFp> vk_s[3], vk_ql, vk_qr, vk_qm, vk_qo, vk_qk;
    // the verifying-key digests are known to be the commitments of THIS key's polynomials under ITS SRS (Setup computed them, or a proof has confirmed it):
    // the digest of the linearised polynomial may then be taken by linearity (seven scalar multiplications on the host) instead of an n-point MSM
    bool vk_consistent = false;
    // per-proof workspace (one proof at a time per key)
    Fr *w_big[5] = {}, *w_small = nullptr;
    std::shared_ptr<std::mutex> mu;
    std::vector<void*> allocs;
    size_t bytes = 0;  // HBM held by the key
};
static std::mutex g_ppk_mu;
static std::map<uint64_t, PlonkPK*> g_ppks;
static uint64_t g_next_ppk = 1;

static const zk_msm_cfg kMont = {0, 1, 0, 0};

static int pk_alloc(PlonkPK* P, Fr** out, size_t elems) {
    void* p = nullptr;
    ZK_HIP(hipMalloc(&p, (elems ? elems : 1) * sizeof(Fr)));
    P->allocs.push_back(p);
    P->bytes += (elems ? elems : 1) * sizeof(Fr);
    *out = (Fr*)p;
    return ZK_OK;
}
static void pk_destroy(PlonkPK* P) {
    if (P->lag_srs) (void)zk_bn254_bases_free(P->lag_srs);
    for (void* p : P->allocs) (void)hipFree(p);
    delete P;
}
static unsigned grid_of(size_t n) { return (unsigned)((n + 255) / 256); }

// kzg.Commit(p, srs) = MultiExp(srs.G1[:len(p)], p), p resident in HBM (Montgomery)
static int commit(const PlonkPK* P, Slot* s, hipStream_t st, const Fr* d_p, size_t len, Affine<HFp>* out) {
    ZK_TRY(slot_sync(s, st));  // the MSM runs on a stream slot of its own
    return zk_bn254_msm_bases_dev(P->srs, 0, d_p, len, &kMont, out);
}

// sum_i k_i P_i for a handful of points on the HOST (digests the prover derives from digests: the linearised polynomial's, the folded quotient's) by the
// interleaved-window method of curve.hpp instead of one double-and-add per term.  The group element, and so the affine result, is the same.
static XYZZ<HFp> host_multi_scalar_mul(const Affine<HFp>* pts, const HFr* ks, int cnt) {
    uint32_t k[8][8];
    if (cnt > 8) cnt = 8;
    for (int t = 0; t < cnt; t++) to_canonical_u32(ks[t], k[t]);
    return multi_scalar_mul(pts, k, cnt);
}

// A commitment in flight on a host thread of its own (the MSM entry point is synchronous and re-entrant: each call takes a stream slot), so that
// independent commitments -- l, r, o; h1, h2, h3; the opening of z next to the linearised polynomial -- and the transforms the main stream keeps
// issuing overlap on the GPU: the bandwidth-bound scalar preparation (digits, sort, task plan) of one MSM runs under the ALU-bound accumulate of another.
struct AsyncCommit {
    std::thread th;
    int rc = ZK_OK;
    std::string err;
    Affine<HFp> out;
    void start(const PlonkPK* P, const Fr* d_p, size_t len) {
        th = std::thread([this, P, d_p, len] {
            rc = zk_bn254_msm_bases_dev(P->srs, 0, d_p, len, &kMont, &out);
            if (rc != ZK_OK) err = zk_last_error();
        });
    }
    int join() {
        if (th.joinable()) th.join();
        return rc == ZK_OK ? ZK_OK : set_err(rc, "%s", err.c_str());
    }
    ~AsyncCommit() { if (th.joinable()) th.join(); }
};
// Three commitments against the same SRS whose polynomials exist at the same moment (l, r, o; h1, h2, h3) as ONE multi-scalar multiplication with three bucket
// sets (msm.hip: k_msm_digits over three scalar vectors, one sort of the 3 x 13 n digits, one task plan, ONE accumulate launch, one reduction per set): three
// concurrent preparations of ~17 dependent launches each, starving each other and whatever the main stream runs meanwhile, become one; the three accumulate kernels
// -- which could not share the machine anyway -- become one launch without seams.  Needs the SRS's window table on ONE device entry; otherwise the thread-per-commit
// path below is used.  Runs on a host thread of its own like AsyncCommit.
struct BatchCommit3 {
    std::thread th;
    int rc = ZK_OK;
    std::string err;
    Affine<HFp> out[3];
    static int run(uint64_t bases, const Fr* const* polys, size_t len, Affine<HFp>* outs, bool wire_values) {
        const void* sc[3] = {polys[0], polys[1], polys[2]};
        static const bool sparse = ZK_EXP("ZKMI_PLONK_SPARSE_DIGITS", 1) != 0;  // 0: wire values sorted with their zero digits (A/B)
        if (wire_values && sparse) return msm_bases_batch_dev_sparse(bases, 0, sc, 3, len, &kMont, outs);
        return zk_bn254_msm_bases_batch_dev(bases, 0, sc, 3, len, &kMont, outs);
    }
    static bool possible(uint64_t bases, const size_t* lens) {
        const void* d_table = nullptr;
        size_t nbases = 0;
        return lens[0] == lens[1] && lens[1] == lens[2] && bases_table(bases, &d_table, nullptr, &nbases) == ZK_OK && d_table && lens[0] <= nbases;
    }
    void start(uint64_t bases, const Fr* const* polys, size_t len, bool wire_values = false) {
        const Fr* p3[3] = {polys[0], polys[1], polys[2]};
        th = std::thread([this, bases, p3, len, wire_values] {
            rc = run(bases, p3, len, out, wire_values);
            if (rc != ZK_OK) err = zk_last_error();
        });
    }
    int join() {
        if (th.joinable()) th.join();
        return rc == ZK_OK ? ZK_OK : set_err(rc, "%s", err.c_str());
    }
    ~BatchCommit3() { if (th.joinable()) th.join(); }
};
static const bool g_plonk_batch3 = ZK_EXP("ZKMI_PLONK_BATCH3", 1) != 0;  // 0: three commitments on three threads (A/B)

// The commitments of one proof as a CHAIN (the schedule of Groth16's msm5): every MSM prepares its scalars (digits, radix sort, task plan --
// bandwidth-bound) on a stream of its own as soon as its polynomial exists, while the accumulate kernels (ALU-bound, each wants the whole
// machine) run ONE AT A TIME in issue order, gated by the previous one's completion event; the reduction tails and the next preparation run
// underneath.  Needs the SRS's window table; otherwise the thread-per-commit fallback above is used.
struct CommitChain {
    static constexpr int NJ = 3;
    const PlonkPK* P = nullptr;
    const void* d_table = nullptr;
    MsmTable tab;
    SlotsGuard<NJ> g;
    MsmPrep prep[NJ];
    MsmJob job[NJ];
    bool live[NJ] = {};
    hipEvent_t last_acc = nullptr;   // completion of the most recently enqueued accumulate (owned by the job that recorded it)
    bool ok = false;

    int init(const PlonkPK* P_, size_t max_len) {
        P = P_;
        size_t n = 0;
        ZK_TRY(bases_table(P->srs, &d_table, &tab, &n));
        if (!d_table) return ZK_OK;  // no table: the caller falls back
        ZK_TRY(acquire_slots(NJ, g.s));
        size_t np = 0, na = 0;
        ZK_TRY(msm_prep_need_table(max_len, tab, g.s[0]->stream, &np, &na, nullptr));
        for (int i = 0; i < NJ; i++) ZK_TRY(g.s[i]->reserve(np + na + 65536));
        ok = true;
        return ZK_OK;
    }
    // d_p must be complete on `producer` (an event is recorded there and awaited by the commit's stream)
    int start(int k, hipStream_t producer, const Fr* d_p, size_t len) {
        Slot* s = g.s[k];
        s->reset();
        hipEvent_t ev;
        ZK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ZK_HIP(hipEventRecord(ev, producer));
        static const bool prep_hi = (ZK_EXP("ZKMI_PLONK_PREP_HI", 1) != 0);  // A/B switch
        hipStream_t sp = prep_hi ? s->hi() : s->stream;  // the preparation at high priority: its sort must get wave slots under a running accumulate
        ZK_TRY(masked_streams(s));
        hipStream_t sa = s->stream;
        if (s->stream_prep) { sp = s->stream_prep; sa = s->stream_acc; }  // experiment ZKMI_CU_SPLIT: disjoint CU sets instead of priorities
        ZK_HIP(hipStreamWaitEvent(sp, ev, 0));
        (void)hipEventDestroy(ev);
        job[k] = MsmJob();
        ZK_TRY(msm_prepare_scalars_table(s, sp, d_p, len, &kMont, tab, &prep[k]));
        live[k] = true;
        job[k].gate_acc = last_acc;
        job[k].want_done = true;
        ZK_TRY(msm_g1_accumulate(s, sa, prep[k], d_table, 0, &job[k]));
        if (job[k].acc_done) last_acc = job[k].acc_done;
        return ZK_OK;
    }
    int finish(int k, Affine<HFp>* out) {
        XYZZ<HFp> t;
        int rc = msm_g1_finish(job[k], &t);
        release(k);
        ZK_TRY(rc);
        *out = t.to_affine();
        return ZK_OK;
    }
    void release(int k) {
        if (!live[k]) return;
        msm_prep_release(&prep[k]);
        if (job[k].acc_done) {
            if (last_acc == job[k].acc_done) last_acc = nullptr;  // the accumulate has completed (finish synchronised): nothing to wait for
            (void)hipEventDestroy(job[k].acc_done);
            job[k].acc_done = nullptr;
        }
        live[k] = false;
    }
    ~CommitChain() {
        for (int k = 0; k < NJ; k++)
            if (live[k]) {
                g.s[k]->sync_hi();
                (void)hipStreamSynchronize(g.s[k]->stream);
                if (g.s[k]->stream_prep) { (void)hipStreamSynchronize(g.s[k]->stream_prep); (void)hipStreamSynchronize(g.s[k]->stream_acc); }
                release(k);
            }
    }
};
// Measured (round 2, 2^22 gates, one box, alternating runs): chain 88.5 - 91.1 ms against 86.8 - 89.7 ms for the thread-per-commit mode, with the preparation
// at normal or high priority and with latency- or work-structured tails -- rocPRIM's onesweep sort does not make progress underneath a running accumulate
// kernel (its look-back tiles spin for wave slots), so "sort under accumulate" buys nothing here and the strict one-at-a-time order costs the overlap the
// threads get by accident.  Off by default; ZKMI_PLONK_CHAIN=1 selects it.
static const bool g_plonk_chain = ZK_EXP("ZKMI_PLONK_CHAIN", 0) == 1;
static const bool g_plonk_serial = ZK_EXP("ZKMI_PLONK_SERIAL", 0) == 1;  // A/B switch: commitments one after the other

// Lagrange (regular) -> canonical (regular) on the small domain, in place: FFTInverse(DIF) + BitReverse, as setup.go / iop.ToCanonical do
static int to_canonical(Slot* s, hipStream_t st, Fr* d, unsigned logn) {
    ZK_TRY(ntt_dev(s, st, d, logn, 1, ZK_DIF, 0));
    return bit_reverse_dev(s, st, d, logn);
}
// canonical p (len coefficients) -> LagrangeCoset on the big domain in `dst` (bit-reversed layout): zero-pad, FFT(DIF, coset)
static int to_big_coset(Slot* s, hipStream_t st, Fr* dst, const Fr* p, size_t len, const PlonkPK* P) {
    if (dst != p) ZK_HIP(hipMemcpyAsync(dst, p, len * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    ZK_HIP(hipMemsetAsync(dst + len, 0, (P->N4 - len) * sizeof(Fr), st));
    return ntt_dev(s, st, dst, P->logN4, 0, ZK_DIF, 1);
}

// scratch carved from the slot arena for the scans over `len` elements
struct ScanBufs {
    uint32_t K = 0, nb = 0;
    Fr *t = nullptr, *b = nullptr, *total = nullptr;
};
static size_t scan_need(size_t len) { return (len / 8 + 4096) * sizeof(Fr) + 8192 + (size_t)8 * (len / 2048 + 2) * sizeof(Fr); }
static int scan_bufs(Slot* s, size_t len, ScanBufs* B) {
    uint32_t K = (uint32_t)((len + 256 * 1024 - 1) / (256 * 1024));
    if (K < 8) K = 8;
    size_t T = (len + K - 1) / K;
    B->K = K;
    B->nb = (uint32_t)((T + 255) / 256);
    B->t = (Fr*)s->alloc((size_t)B->nb * 256 * sizeof(Fr));
    B->b = (Fr*)s->alloc((size_t)HSCAN_BATCH_MAX * B->nb * sizeof(Fr) + 64);  // HSCAN_BATCH_MAX rows: poly_eval_batch_dev
    B->total = (Fr*)s->alloc(64);
    if (!B->t || !B->b || !B->total) return set_err(ZK_ERR_HIP, "PLONK scan workspace was not reserved");
    return ZK_OK;
}
// f_k(a) -> d_out[k] for cnt <= HSCAN_BATCH_MAX polynomials, two launches in all
static int poly_eval_batch_dev(Slot* s, hipStream_t st, const Fr* const* f, const size_t* len, int cnt, const HFr& a, const ScanBufs& B, Fr* d_out) {
    if (cnt < 1 || cnt > HSCAN_BATCH_MAX) return set_err(ZK_ERR_ARG, "evaluation batch of %d", cnt);
    HFr A = HFr::one(), M;
    {
        HFr base = a;
        for (uint32_t k = B.K; k; k >>= 1) { if (k & 1) A = A * base; base = base.sqr(); }
        M = A;
        for (int i = 0; i < 8; i++) M = M.sqr();
    }
    HscanBatch Bt;
    for (int k = 0; k < HSCAN_BATCH_MAX; k++) { Bt.f[k] = f[k < cnt ? k : 0]; Bt.len[k] = len[k < cnt ? k : 0]; }
    ZK_LAUNCH(s, st, "plonk_horner_local", k_hscan_local_batch, dim3(B.nb, (unsigned)cnt), dim3(256), 0, Bt, B.K, to_dev(a), to_dev(A), B.nb, B.b);
    ZK_LAUNCH(s, st, "plonk_horner_blocks", k_hscan_blocks, dim3((unsigned)cnt), dim3(1024), 0, B.b, B.nb, to_dev(M), d_out);
    return ZK_OK;
}
// q = (f - f(a)) / (X - a) (len - 1 coefficients; q may alias f; q[len-1] is set to 0), f(a) -> *d_eval
static int poly_divide_dev(Slot* s, hipStream_t st, const Fr* f, size_t len, const HFr& a, const ScanBufs& B, Fr* q, Fr* d_eval) {
    HFr A = HFr::one(), M;
    {
        HFr base = a;
        for (uint32_t k = B.K; k; k >>= 1) { if (k & 1) A = A * base; base = base.sqr(); }
        M = A;
        for (int i = 0; i < 8; i++) M = M.sqr();
    }
    ZK_LAUNCH(s, st, "plonk_horner_local", k_hscan_local, dim3(B.nb), dim3(256), 0, f, len, B.K, to_dev(a), to_dev(A), B.t, B.b);
    ZK_LAUNCH(s, st, "plonk_horner_blocks", k_hscan_blocks, dim3(1), dim3(1024), 0, B.b, B.nb, to_dev(M), d_eval);
    ZK_LAUNCH(s, st, "plonk_divide_apply", k_hscan_apply, dim3(B.nb), dim3(256), 0, f, len, B.K, to_dev(a), to_dev(A), (const Fr*)B.t, (const Fr*)B.b, q);
    return ZK_OK;
}

// sigma in Lagrange form from the permutation (what BuildRatioCopyConstraint evaluates through pk.Permutation)
static int make_sigma(PlonkPK* P, Slot* s, hipStream_t st, const uint32_t* d_perm) {
    const size_t n = P->n;
    Domain* d0;
    ZK_TRY(get_domain(s, st, P->logn, DOM_TW, &d0));
    P->gen = d0->gen;
    P->u = d0->coset;
    P->card_inv = d0->card_inv;
    ZK_TRY(pk_alloc(P, &P->sig, 3 * n));
    ZK_LAUNCH(s, st, "plonk_sigma_lagrange", k_sigma_lagrange, dim3(grid_of(3 * n)), dim3(256), 0, d_perm, (const Fr*)d0->tw, (uint32_t)n, to_dev(P->u),
              to_dev(P->u * P->u), P->sig);
    return ZK_OK;
}
// From the canonical polynomials (all in P) to a usable key: the nine big-coset tables and the per-proof workspace.
static int finish_pk(PlonkPK* P, Slot* s, hipStream_t st) {
    const size_t n = P->n, N4 = P->N4;
    const Fr* canon[7] = {P->ql, P->qr, P->qm, P->qo, P->s1, P->s2, P->s3};
    for (int k = 0; k < 9; k++) ZK_TRY(pk_alloc(P, &P->e[k], N4));
    for (int k = 0; k < 7; k++) ZK_TRY(to_big_coset(s, st, P->e[k], canon[k], n, P));
    // L_1 = (X^n - 1) / (n (X - 1)) = (1/n) (1 + X + ... + X^(n-1))
    ZK_LAUNCH(s, st, "plonk_fill", k_fill, dim3(grid_of(n)), dim3(256), 0, P->e[7], n, to_dev(P->card_inv));
    ZK_TRY(to_big_coset(s, st, P->e[7], P->e[7], n, P));
    // id = X on the coset: g * W^bitrev(i)
    ZK_HIP(hipMemsetAsync(P->e[8], 0, N4 * sizeof(Fr), st));
    {
        Fr one = Fr::one();
        ZK_HIP(hipMemcpyAsync(P->e[8] + 1, &one, sizeof(Fr), hipMemcpyHostToDevice, st));
        ZK_HIP(hipStreamSynchronize(st));
    }
    ZK_TRY(ntt_dev(s, st, P->e[8], P->logN4, 0, ZK_DIF, 1));
    ZK_TRY(pk_alloc(P, &P->e_cqk, N4));
    ZK_TRY(to_big_coset(s, st, P->e_cqk, P->cqk, n, P));
    for (int k = 0; k < 5; k++) ZK_TRY(pk_alloc(P, &P->w_big[k], N4));
    ZK_TRY(pk_alloc(P, &P->w_small, 16 * (n + 8)));
    P->mu = std::make_shared<std::mutex>();
    return slot_sync(s, st);
}

static int check_srs(uint64_t srs, size_t n) {
    size_t cnt = 0;
    int g2 = 0;
    ZK_TRY(bases_info(srs, &cnt, &g2));
    if (g2) return set_err(ZK_ERR_ARG, "the KZG SRS must be a G1 base array");
    if (cnt < n + 3) return set_err(ZK_ERR_ARG, "kzg: invalid polynomial size: the SRS holds %zu points, %zu needed (domain + 3)", cnt, n + 3);
    return ZK_OK;
}

// setup.go buildPermutation, on the host (a sequential pass over the 3n wire slots)
static void build_permutation(size_t n, size_t npub, size_t nc, size_t nvars, const uint32_t* xa, const uint32_t* xb, const uint32_t* xc, std::vector<uint32_t>* perm) {
    std::vector<uint32_t> lro(3 * n, 0);
    for (size_t i = 0; i < npub; i++) lro[i] = (uint32_t)i;
    for (size_t i = 0; i < nc; i++) {
        lro[npub + i] = xa[i];
        lro[n + npub + i] = xb[i];
        lro[2 * n + npub + i] = xc[i];
    }
    const int64_t none = -1;
    std::vector<int64_t> cycle(nvars ? nvars : 1, none), pm(3 * n, none);
    for (size_t i = 0; i < 3 * n; i++) {
        if (cycle[lro[i]] != none) pm[i] = cycle[lro[i]];
        cycle[lro[i]] = (int64_t)i;
    }
    perm->resize(3 * n);
    for (size_t i = 0; i < 3 * n; i++) (*perm)[i] = (uint32_t)(pm[i] == none ? cycle[lro[i]] : pm[i]);
}

static int register_pk(PlonkPK* P, uint64_t* handle) {
    std::lock_guard<std::mutex> lk(g_ppk_mu);
    *handle = hmake(g_next_ppk++);
    g_ppks[*handle] = P;
    return ZK_OK;
}

// the count / domain rules live with the other readers of untrusted bytes (text_host.hpp: host only, sanitizer- and mutation-tested)
static int check_counts(uint64_t n_public, size_t n_vars, size_t n_constraints) {
    std::string e;
    const int rc = plonk_check_counts(n_public, n_vars, n_constraints, &e);
    return rc == ZK_OK ? ZK_OK : set_err(rc, "%s", e.c_str());
}
static int domains_for(size_t size_system, unsigned* logn, unsigned* logN4) {
    std::string e;
    const int rc = plonk_domains_for(size_system, logn, logN4, &e);
    return rc == ZK_OK ? ZK_OK : set_err(rc, "%s", e.c_str());
}

}  // namespace zkmi

using namespace zkmi;

extern "C" {

int zk_bn254_plonk_synth_qk_dev(void* d_qk, const void* d_ql, const void* d_qr, const void* d_qo, const void* d_qm, const void* d_xa, const void* d_xb,
                                const void* d_xc, const void* d_solution, size_t nc, void* stream) {
    if (nc && (!d_qk || !d_ql || !d_qr || !d_qo || !d_qm || !d_xa || !d_xb || !d_xc || !d_solution)) return set_err(ZK_ERR_ARG, "null pointer");
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = stream ? (hipStream_t)stream : g.s->stream;
    if (nc)
        ZK_LAUNCH(g.s, st, "plonk_synth_qk", k_synth_qk, dim3(grid_of(nc)), dim3(256), 0, (Fr*)d_qk, (const Fr*)d_ql, (const Fr*)d_qr, (const Fr*)d_qo, (const Fr*)d_qm,
                  (const uint32_t*)d_xa, (const uint32_t*)d_xb, (const uint32_t*)d_xc, (const Fr*)d_solution, nc);
    return stream ? ZK_OK : slot_sync(g.s, st);
}

int zk_bn254_plonk_setup(const zk_plonk_circuit* c, uint64_t srs, uint64_t* handle, zk_plonk_vk* vk) {
    ZK_ON_ENTRY_OF(srs);
    if (!c || !handle) return set_err(ZK_ERR_ARG, "null pointer");
    const size_t npub = c->n_public, nc = c->n_constraints;
    if (nc && (!c->ql || !c->qr || !c->qo || !c->qm || !c->qk || !c->xa || !c->xb || !c->xc)) return set_err(ZK_ERR_ARG, "null pointer");
    if (npub > c->n_vars) return set_err(ZK_ERR_ARG, "more public inputs than variables");
    for (size_t i = 0; i < nc; i++)
        if (c->xa[i] >= c->n_vars || c->xb[i] >= c->n_vars || c->xc[i] >= c->n_vars) return set_err(ZK_ERR_ARG, "gate %zu names a wire outside the %zu variables", i, c->n_vars);
    unsigned logn, logN4;
    ZK_TRY(domains_for(nc + npub, &logn, &logN4));
    ZK_TRY(ensure_init());
    ZK_TRY(check_srs(srs, (size_t)1 << logn));
    std::unique_ptr<PlonkPK, void (*)(PlonkPK*)> P(new PlonkPK(), pk_destroy);
    P->logn = logn; P->logN4 = logN4; P->log_rho = logN4 - logn;
    P->n = (size_t)1 << logn; P->N4 = (size_t)1 << logN4;
    P->n_public = npub; P->n_constraints = nc; P->n_vars = c->n_vars;
    P->srs = srs;
    const size_t n = P->n;
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    Slot* s = g.s;
    hipStream_t st = s->stream;
    ZK_TRY(s->reserve(5 * nc * sizeof(Fr) + 3 * n * 4 + 65536));
    // selectors: Lagrange form [placeholders | gates | 0], then canonical
    Fr** dst[6] = {&P->ql, &P->qr, &P->qm, &P->qo, &P->cqk, &P->lqk};
    const void* src[6] = {c->ql, c->qr, c->qm, c->qo, c->qk, c->qk};
    const Fr minus_one = to_dev(HFr::zero() - HFr::one());
    Fr* stage[5] = {};
    if (!c->coeffs_on_device)
        for (int k = 0; k < 5; k++) {
            stage[k] = (Fr*)s->alloc(nc * sizeof(Fr) + 16);
            if (nc) ZK_HIP(hipMemcpyAsync(stage[k], src[k], nc * sizeof(Fr), hipMemcpyHostToDevice, st));
        }
    for (int k = 0; k < 6; k++) {
        ZK_TRY(pk_alloc(P.get(), dst[k], n));
        const Fr* from = c->coeffs_on_device ? (const Fr*)src[k] : stage[k < 5 ? k : 4];
        ZK_LAUNCH(s, st, "plonk_place_selector", k_place_selector, dim3(grid_of(n)), dim3(256), 0, *dst[k], from, (uint32_t)npub, (uint32_t)nc, (uint32_t)n,
                  k == 0 ? minus_one : Fr::zero());
        if (k < 5) ZK_TRY(to_canonical(s, st, *dst[k], logn));  // LQk stays in Lagrange form (the prover completes it with the public inputs)
    }
    // wire ids (prove's gather) and the permutation
    uint32_t** wid[3] = {&P->xa, &P->xb, &P->xc};
    const uint32_t* wsrc[3] = {c->xa, c->xb, c->xc};
    for (int k = 0; k < 3; k++) {
        Fr* tmp;
        ZK_TRY(pk_alloc(P.get(), &tmp, (nc * 4 + 31) / 32 + 1));
        *wid[k] = (uint32_t*)tmp;
        if (nc) ZK_HIP(hipMemcpyAsync(*wid[k], wsrc[k], nc * 4, hipMemcpyHostToDevice, st));
    }
    std::vector<uint32_t> perm;
    build_permutation(n, npub, nc, c->n_vars, c->xa, c->xb, c->xc, &perm);
    uint32_t* d_perm;
    {
        Fr* tmp;
        ZK_TRY(pk_alloc(P.get(), &tmp, (3 * n * 4 + 31) / 32 + 1));
        P->perm = d_perm = (uint32_t*)tmp;
    }
    ZK_HIP(hipMemcpyAsync(d_perm, perm.data(), 3 * n * 4, hipMemcpyHostToDevice, st));
    ZK_TRY(pk_alloc(P.get(), &P->s1, n));
    ZK_TRY(pk_alloc(P.get(), &P->s2, n));
    ZK_TRY(pk_alloc(P.get(), &P->s3, n));
    ZK_TRY(make_sigma(P.get(), s, st, d_perm));
    Fr* sc[3] = {P->s1, P->s2, P->s3};
    for (int k = 0; k < 3; k++) {  // S1..S3 canonical
        ZK_HIP(hipMemcpyAsync(sc[k], P->sig + k * n, n * sizeof(Fr), hipMemcpyDeviceToDevice, st));
        ZK_TRY(to_canonical(s, st, sc[k], logn));
    }
    ZK_TRY(finish_pk(P.get(), s, st));
    // verifying key: commitments to the canonical polynomials
    const Fr* cm[8] = {P->s1, P->s2, P->s3, P->ql, P->qr, P->qm, P->qo, P->cqk};
    Affine<HFp>* cmo[8] = {&P->vk_s[0], &P->vk_s[1], &P->vk_s[2], &P->vk_ql, &P->vk_qr, &P->vk_qm, &P->vk_qo, &P->vk_qk};
    for (int k = 0; k < 8; k++) ZK_TRY(commit(P.get(), s, st, cm[k], n, cmo[k]));
    P->vk_consistent = true;
    if (vk) {
        memset(vk, 0, sizeof *vk);
        vk->size = n;
        vk->n_public = npub;
        memcpy(&vk->size_inv, &P->card_inv, 32);
        memcpy(&vk->generator, &P->gen, 32);
        memcpy(&vk->coset_shift, &P->u, 32);
        memcpy(vk->s, P->vk_s, 3 * 64);
        memcpy(&vk->ql, &P->vk_ql, 64); memcpy(&vk->qr, &P->vk_qr, 64); memcpy(&vk->qm, &P->vk_qm, 64);
        memcpy(&vk->qo, &P->vk_qo, 64); memcpy(&vk->qk, &P->vk_qk, 64);
    }
    return register_pk(P.release(), handle);
}

int zk_bn254_plonk_pk_load(const zk_plonk_pk* k, uint64_t srs, uint64_t* handle) {
    ZK_ON_ENTRY_OF(srs);
    if (!k || !handle) return set_err(ZK_ERR_ARG, "null pointer");
    if (!k->ql || !k->qr || !k->qm || !k->qo || !k->cqk || !k->lqk || !k->s1 || !k->s2 || !k->s3 || !k->permutation || !k->vk_s || !k->vk_ql || !k->vk_qr ||
        !k->vk_qm || !k->vk_qo || !k->vk_qk || (k->n_constraints && (!k->xa || !k->xb || !k->xc)))
        return set_err(ZK_ERR_ARG, "null pointer");
    unsigned logn, logN4;
    ZK_TRY(check_counts(k->n_public, k->n_vars, k->n_constraints));
    ZK_TRY(domains_for(k->n_constraints + k->n_public, &logn, &logN4));
    if (logn != k->log_n) return set_err(ZK_ERR_ARG, "log_n = %u does not match %zu constraints + %zu public inputs", k->log_n, k->n_constraints, k->n_public);
    const size_t n = (size_t)1 << logn;
    for (size_t i = 0; i < k->n_constraints; i++)
        if (k->xa[i] >= k->n_vars || k->xb[i] >= k->n_vars || k->xc[i] >= k->n_vars) return set_err(ZK_ERR_ARG, "gate %zu names a wire outside the %zu variables", i, k->n_vars);
    std::vector<uint32_t> perm(3 * n);
    for (size_t i = 0; i < 3 * n; i++) {
        if (k->permutation[i] < 0 || (size_t)k->permutation[i] >= 3 * n) return set_err(ZK_ERR_ARG, "Permutation[%zu] out of range", i);
        perm[i] = (uint32_t)k->permutation[i];
    }
    ZK_TRY(ensure_init());
    ZK_TRY(check_srs(srs, n));
    std::unique_ptr<PlonkPK, void (*)(PlonkPK*)> P(new PlonkPK(), pk_destroy);
    P->logn = logn; P->logN4 = logN4; P->log_rho = logN4 - logn;
    P->n = n; P->N4 = (size_t)1 << logN4;
    P->n_public = k->n_public; P->n_constraints = k->n_constraints; P->n_vars = k->n_vars;
    P->srs = srs;
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    Slot* s = g.s;
    hipStream_t st = s->stream;
    ZK_TRY(s->reserve(3 * n * 4 + 65536));
    Fr** dst[9] = {&P->ql, &P->qr, &P->qm, &P->qo, &P->cqk, &P->lqk, &P->s1, &P->s2, &P->s3};
    const zk_fr* src[9] = {k->ql, k->qr, k->qm, k->qo, k->cqk, k->lqk, k->s1, k->s2, k->s3};
    for (int i = 0; i < 9; i++) {
        ZK_TRY(pk_alloc(P.get(), dst[i], n));
        ZK_HIP(hipMemcpyAsync(*dst[i], src[i], n * sizeof(Fr), hipMemcpyHostToDevice, st));
    }
    uint32_t** wid[3] = {&P->xa, &P->xb, &P->xc};
    const uint32_t* wsrc[3] = {k->xa, k->xb, k->xc};
    for (int i = 0; i < 3; i++) {
        Fr* tmp;
        ZK_TRY(pk_alloc(P.get(), &tmp, (k->n_constraints * 4 + 31) / 32 + 1));
        *wid[i] = (uint32_t*)tmp;
        if (k->n_constraints) ZK_HIP(hipMemcpyAsync(*wid[i], wsrc[i], k->n_constraints * 4, hipMemcpyHostToDevice, st));
    }
    uint32_t* d_perm;
    {
        Fr* tmp;
        ZK_TRY(pk_alloc(P.get(), &tmp, (3 * n * 4 + 31) / 32 + 1));
        P->perm = d_perm = (uint32_t*)tmp;
    }
    ZK_HIP(hipMemcpyAsync(d_perm, perm.data(), 3 * n * 4, hipMemcpyHostToDevice, st));
    ZK_TRY(make_sigma(P.get(), s, st, d_perm));
    ZK_TRY(finish_pk(P.get(), s, st));
    memcpy(P->vk_s, k->vk_s, 3 * 64);
    memcpy(&P->vk_ql, k->vk_ql, 64); memcpy(&P->vk_qr, k->vk_qr, 64); memcpy(&P->vk_qm, k->vk_qm, 64);
    memcpy(&P->vk_qo, k->vk_qo, 64); memcpy(&P->vk_qk, k->vk_qk, 64);
    return register_pk(P.release(), handle);
}

// ---- plonk.ProvingKey.ReadFrom / WriteTo (gnark v0.8.0 internal/backend/bn254/plonk/marshal.go  [UPSTREAM-RECALL]; the reference ships the key as
// hex of these bytes: internal/backend/helpers.go:49-60,82-87, produced at main.go:58-78, consumed at main.go:24-37):
//   VerifyingKey.WriteTo  Size u64 | SizeInv | Generator | NbPublicVariables u64 | CosetShift | S[0..2] | Ql Qr Qm Qo Qk (compressed G1)   368 B
//   Domain[0], Domain[1]  Cardinality u64 | CardinalityInv | Generator | GeneratorInv | FrMultiplicativeGen | FrMultiplicativeGenInv      168 B each
//   Ql Qr Qm Qo CQk LQk S1Canonical S2Canonical S3Canonical   each u32 BE length | n x 32 B BE
//   Permutation           3n raw big-endian int64
static const size_t PK_HEAD = PLONK_PK_HEAD;

int zk_bn254_plonk_pk_read(const void* data, size_t len, int is_hex, size_t n_vars, size_t n_constraints, const uint32_t* xa, const uint32_t* xb,
                           const uint32_t* xc, uint64_t srs, uint64_t* handle) {
    ZK_ON_ENTRY_OF(srs);
    if (!data || !handle || (n_constraints && (!xa || !xb || !xc))) return set_err(ZK_ERR_ARG, "null pointer");
    PlonkKeyHeader H;  // every size below comes out of this host-side reading of the header and the nine length prefixes (text_host.hpp)
    {
        std::string e;
        const int rc = plonk_pk_header(data, len, is_hex, n_vars, n_constraints, &H, &e);
        if (rc != ZK_OK) return set_err(rc, "%s", e.c_str());
    }
    const size_t n = H.n, nbytes = H.nbytes;
    const uint64_t npub = H.n_public;
    const unsigned logn = H.logn, logN4 = H.logN4;
    const auto t_start = std::chrono::steady_clock::now();
    for (size_t i = 0; i < n_constraints; i++)
        if (xa[i] >= n_vars || xb[i] >= n_vars || xc[i] >= n_vars) return set_err(ZK_ERR_ARG, "gate %zu names a wire outside the %zu variables", i, n_vars);
    ZK_TRY(ensure_init());
    ZK_TRY(check_srs(srs, n));
    std::unique_ptr<PlonkPK, void (*)(PlonkPK*)> P(new PlonkPK(), pk_destroy);
    P->logn = logn; P->logN4 = logN4; P->log_rho = logN4 - logn;
    P->n = n; P->N4 = (size_t)1 << logN4;
    P->n_public = (size_t)npub; P->n_constraints = n_constraints; P->n_vars = n_vars;
    P->srs = srs;
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    Slot* s = g.s;
    hipStream_t st = s->stream;
    ZK_TRY(s->reserve(len + nbytes + 65536));
    int* d_status = (int*)s->alloc(64);
    ZK_HIP(hipMemsetAsync(d_status, 0, 4, st));
    uint8_t* d_bytes = (uint8_t*)s->alloc(nbytes + 16);
    if (is_hex) {
        void* d_text = s->alloc(len + 16);
        ZK_TRY(h2d_big(d_text, data, len, st));
        ZK_TRY(hex_decode_dev(s, st, d_text, nbytes, d_bytes, d_status));
    } else {
        ZK_TRY(h2d_big(d_bytes, data, nbytes, st));
    }
    Fr** dst[9] = {&P->ql, &P->qr, &P->qm, &P->qo, &P->cqk, &P->lqk, &P->s1, &P->s2, &P->s3};
    for (int k = 0; k < 9; k++) {
        ZK_TRY(pk_alloc(P.get(), dst[k], n));
        ZK_TRY(fr_from_be_dev(s, st, d_bytes + PK_HEAD + (size_t)k * (4 + 32 * n) + 4, n, *dst[k], d_status));
    }
    {
        Fr* tmp;
        ZK_TRY(pk_alloc(P.get(), &tmp, (3 * n * 4 + 31) / 32 + 1));
        P->perm = (uint32_t*)tmp;
        ZK_LAUNCH(s, st, "plonk_perm_from_be", k_perm_from_be, dim3(grid_of(3 * n)), dim3(256), 0, (const uint32_t*)(d_bytes + PK_HEAD + 9 * (4 + 32 * n)), 3 * n, P->perm, d_status);
    }
    // the verifying key's eight digests: compressed G1 at bytes 112 .. 368
    Affine<Fp>* d_vk = (Affine<Fp>*)s->alloc(8 * 64 + 16);
    ZK_TRY(g1_decompress_dev(s, st, d_bytes + 112, 8, d_vk, d_status));
    Affine<HFp> vkp[8];
    int h_status = 0;
    ZK_HIP(hipMemcpyAsync(vkp, d_vk, sizeof vkp, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipMemcpyAsync(&h_status, d_status, 4, hipMemcpyDeviceToHost, st));
    ZK_TRY(slot_sync(s, st));
    if (h_status & 1) return set_err(ZK_ERR_ARG, "proving key: invalid hex character");
    if (h_status & 2) return set_err(ZK_ERR_ARG, "proving key: invalid fr.Element encoding (value >= r)");
    if (h_status & 4) return set_err(ZK_ERR_ARG, "proving key: invalid compressed G1 point in the verifying key");
    if (h_status & 8) return set_err(ZK_ERR_ARG, "proving key: Permutation entry out of range");
    memcpy(P->vk_s, vkp, 3 * 64);
    P->vk_ql = vkp[3]; P->vk_qr = vkp[4]; P->vk_qm = vkp[5]; P->vk_qo = vkp[6]; P->vk_qk = vkp[7];
    uint32_t** wid[3] = {&P->xa, &P->xb, &P->xc};
    const uint32_t* wsrc[3] = {xa, xb, xc};
    for (int i = 0; i < 3; i++) {
        Fr* tmp;
        ZK_TRY(pk_alloc(P.get(), &tmp, (n_constraints * 4 + 31) / 32 + 1));
        *wid[i] = (uint32_t*)tmp;
        if (n_constraints) ZK_HIP(hipMemcpyAsync(*wid[i], wsrc[i], n_constraints * 4, hipMemcpyHostToDevice, st));
    }
    const auto t_decoded = std::chrono::steady_clock::now();
    ZK_TRY(make_sigma(P.get(), s, st, P->perm));
    ZK_TRY(finish_pk(P.get(), s, st));
    prof_host("export.pk_text_to_device", std::chrono::duration<double, std::milli>(t_decoded - t_start).count());
    prof_host("export.pk_coset_forms", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_decoded).count());
    return register_pk(P.release(), handle);
}

int zk_bn254_plonk_pk_write(uint64_t handle, int as_hex, void* out, size_t cap, size_t* out_len) {
    ZK_ON_ENTRY_OF(handle);
    if (!out || !out_len) return set_err(ZK_ERR_ARG, "null pointer");
    PlonkPK* P;
    std::shared_ptr<std::mutex> mu;
    {
        std::lock_guard<std::mutex> lk(g_ppk_mu);
        auto it = g_ppks.find(handle);
        if (it == g_ppks.end()) return set_err(ZK_ERR_HANDLE, "unknown PLONK proving-key handle %llu", (unsigned long long)handle);
        P = it->second;
        mu = P->mu;
    }
    std::lock_guard<std::mutex> key_lock(*mu);
    {
        std::lock_guard<std::mutex> lk(g_ppk_mu);
        auto it = g_ppks.find(handle);
        if (it == g_ppks.end() || it->second != P) return set_err(ZK_ERR_HANDLE, "PLONK proving key %llu was freed", (unsigned long long)handle);
    }
    const size_t n = P->n, nbytes = PK_HEAD + 9 * (4 + 32 * n) + 24 * n, need = as_hex ? 2 * nbytes : nbytes;
    *out_len = need;
    if (cap < need) return set_err(ZK_ERR_ARG, "output holds %zu bytes, %zu needed", cap, need);
    uint8_t head[PK_HEAD];
    memset(head, 0, sizeof head);
    auto put64 = [](uint8_t* p, uint64_t v) { for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (56 - 8 * i)); };
    put64(head, n);
    fr_to_be(P->card_inv, head + 8);
    fr_to_be(P->gen, head + 40);
    put64(head + 72, P->n_public);
    fr_to_be(P->u, head + 80);
    const Affine<HFp>* vkp[8] = {&P->vk_s[0], &P->vk_s[1], &P->vk_s[2], &P->vk_ql, &P->vk_qr, &P->vk_qm, &P->vk_qo, &P->vk_qk};
    for (int k = 0; k < 8; k++) g1_compress(*vkp[k], head + 112 + 32 * k);
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    Slot* s = g.s;
    hipStream_t st = s->stream;
    Domain *d0, *d1;
    ZK_TRY(get_domain(s, st, P->logn, 0, &d0));
    ZK_TRY(get_domain(s, st, P->logN4, 0, &d1));
    Domain* dd[2] = {d0, d1};
    for (int k = 0; k < 2; k++) {
        uint8_t* o = head + 368 + 168 * k;
        put64(o, (uint64_t)1 << dd[k]->logn);
        fr_to_be(dd[k]->card_inv, o + 8);
        fr_to_be(dd[k]->gen, o + 40);
        fr_to_be(dd[k]->gen_inv, o + 72);
        fr_to_be(dd[k]->coset, o + 104);
        fr_to_be(dd[k]->coset_inv, o + 136);
    }
    ZK_TRY(s->reserve(3 * nbytes + 65536));
    uint8_t* d_bytes = (uint8_t*)s->alloc(nbytes + 16);
    ZK_HIP(hipMemcpyAsync(d_bytes, head, PK_HEAD, hipMemcpyHostToDevice, st));
    const Fr* src[9] = {P->ql, P->qr, P->qm, P->qo, P->cqk, P->lqk, P->s1, P->s2, P->s3};
    uint8_t pre[4] = {(uint8_t)(n >> 24), (uint8_t)(n >> 16), (uint8_t)(n >> 8), (uint8_t)n};
    for (int k = 0; k < 9; k++) {
        uint8_t* o = d_bytes + PK_HEAD + (size_t)k * (4 + 32 * n);
        ZK_HIP(hipMemcpyAsync(o, pre, 4, hipMemcpyHostToDevice, st));
        ZK_TRY(fr_to_be_dev(s, st, src[k], n, o + 4));
    }
    ZK_HIP(hipStreamSynchronize(st));  // `pre` and `head` live on this stack frame
    ZK_LAUNCH(s, st, "plonk_perm_to_be", k_perm_to_be, dim3(grid_of(3 * n)), dim3(256), 0, (const uint32_t*)P->perm, 3 * n, (uint32_t*)(d_bytes + PK_HEAD + 9 * (4 + 32 * n)));
    if (as_hex) {
        void* d_text = s->alloc(2 * nbytes + 16);
        ZK_TRY(hex_encode_dev(s, st, d_bytes, nbytes, d_text));
        ZK_HIP(hipMemcpyAsync(out, d_text, 2 * nbytes, hipMemcpyDeviceToHost, st));
    } else {
        ZK_HIP(hipMemcpyAsync(out, d_bytes, nbytes, hipMemcpyDeviceToHost, st));
    }
    return slot_sync(s, st);
}

int zk_bn254_plonk_pk_info(uint64_t handle, size_t* domain_size, size_t* n_public, size_t* n_constraints, size_t* n_vars) {
    ZK_ON_ENTRY_OF(handle);
    std::lock_guard<std::mutex> lk(g_ppk_mu);
    auto it = g_ppks.find(handle);
    if (it == g_ppks.end()) return set_err(ZK_ERR_HANDLE, "unknown PLONK proving key %llu", (unsigned long long)handle);
    const PlonkPK* P = it->second;
    if (domain_size) *domain_size = P->n;
    if (n_public) *n_public = P->n_public;
    if (n_constraints) *n_constraints = P->n_constraints;
    if (n_vars) *n_vars = P->n_vars;
    return ZK_OK;
}

int zk_bn254_plonk_pk_bytes(uint64_t handle, size_t* bytes) {
    ZK_ON_ENTRY_OF(handle);
    if (!bytes) return set_err(ZK_ERR_ARG, "null pointer");
    std::lock_guard<std::mutex> lk(g_ppk_mu);
    auto it = g_ppks.find(handle);
    if (it == g_ppks.end()) return set_err(ZK_ERR_HANDLE, "unknown PLONK proving key %llu", (unsigned long long)handle);
    *bytes = it->second->bytes;
    return ZK_OK;
}

// The SRS in Lagrange form over this key's domain, built once (lagrange.hip: n/2 log2 n + n point-by-scalar multiplications -- 0.15 s at 2^19 gates, 1.3 s at 2^22):
// from then on zk_bn254_plonk_prove commits l, r, o from the wire values (same digests, a quarter to a third of the additions for the scalars of a real
// circuit).  For a prover that keeps its key; needs the SRS on one device entry.  A key that has it: ZK_OK, untouched.
int zk_bn254_plonk_pk_lagrange_srs(uint64_t handle) {
    ZK_ON_ENTRY_OF(handle);
    PlonkPK* P;
    std::shared_ptr<std::mutex> mu;
    {
        std::lock_guard<std::mutex> lk(g_ppk_mu);
        auto it = g_ppks.find(handle);
        if (it == g_ppks.end()) return set_err(ZK_ERR_HANDLE, "unknown PLONK proving-key handle %llu", (unsigned long long)handle);
        P = it->second;
        mu = P->mu;
    }
    std::lock_guard<std::mutex> work(*mu);
    {
        std::lock_guard<std::mutex> lk(g_ppk_mu);
        auto it = g_ppks.find(handle);
        if (it == g_ppks.end() || it->second != P) return set_err(ZK_ERR_HANDLE, "PLONK proving key %llu was freed", (unsigned long long)handle);
    }
    if (P->lag_srs) return ZK_OK;
    uint64_t h = 0;
    ZK_TRY(zk_bn254_bases_lagrange(P->srs, P->logn, &h));
    P->lag_srs = h;
    const void* d_table = nullptr;
    MsmTable tab;
    size_t nb = 0;
    if (bases_table(h, &d_table, &tab, &nb) == ZK_OK) P->bytes += (P->n + 2) * 64 * (1 + (d_table ? (size_t)tab.rows() : 0));
    return ZK_OK;
}

int zk_bn254_plonk_pk_free(uint64_t handle) {
    ZK_ON_ENTRY_OF(handle);
    PlonkPK* P;
    {
        std::lock_guard<std::mutex> lk(g_ppk_mu);
        auto it = g_ppks.find(handle);
        if (it == g_ppks.end()) return set_err(ZK_ERR_HANDLE, "unknown PLONK proving-key handle %llu", (unsigned long long)handle);
        P = it->second;
        g_ppks.erase(it);
    }
    { std::lock_guard<std::mutex> lk(*P->mu); }  // a proof in flight finishes first
    pk_destroy(P);
    return ZK_OK;
}

int zk_bn254_plonk_pk_export(uint64_t handle, int which, zk_fr* out, size_t cnt) {
    ZK_ON_ENTRY_OF(handle);
    if (!out) return set_err(ZK_ERR_ARG, "null pointer");
    std::lock_guard<std::mutex> lk(g_ppk_mu);
    auto it = g_ppks.find(handle);
    if (it == g_ppks.end()) return set_err(ZK_ERR_HANDLE, "unknown PLONK proving-key handle %llu", (unsigned long long)handle);
    PlonkPK* P = it->second;
    const Fr* src[9] = {P->ql, P->qr, P->qm, P->qo, P->cqk, P->s1, P->s2, P->s3, P->lqk};
    if (which < 0 || which > 8 || cnt > P->n) return set_err(ZK_ERR_ARG, "bad polynomial index / length");
    ZK_HIP(hipMemcpy(out, src[which], cnt * sizeof(Fr), hipMemcpyDeviceToHost));
    return ZK_OK;
}

int zk_bn254_plonk_prove(uint64_t handle, const void* solution, size_t n_vars, int on_device, const zk_fr blinders[9], const zk_fr* challenges,
                         uint8_t proof_out[ZK_PLONK_PROOF_BYTES]) {
    ZK_ON_ENTRY_OF(handle);
    if (!solution || !blinders || !proof_out) return set_err(ZK_ERR_ARG, "null pointer");
    PlonkPK* P;
    std::shared_ptr<std::mutex> mu;
    {
        std::lock_guard<std::mutex> lk(g_ppk_mu);
        auto it = g_ppks.find(handle);
        if (it == g_ppks.end()) return set_err(ZK_ERR_HANDLE, "unknown PLONK proving-key handle %llu", (unsigned long long)handle);
        P = it->second;
        mu = P->mu;
    }
    std::lock_guard<std::mutex> proof_lock(*mu);
    {   // the key may have been freed between the lookup and the lock (pk_free waits on this mutex and then destroys it)
        std::lock_guard<std::mutex> lk(g_ppk_mu);
        auto it = g_ppks.find(handle);
        if (it == g_ppks.end() || it->second != P) return set_err(ZK_ERR_HANDLE, "PLONK proving key %llu was freed", (unsigned long long)handle);
    }
    if (n_vars != P->n_vars) return set_err(ZK_ERR_LEN, "len(solution) = %zu != %zu variables of the constraint system", n_vars, P->n_vars);
    const size_t n = P->n, N4 = P->N4, npub = P->n_public;
    const unsigned logn = P->logn, logN4 = P->logN4;
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    Slot* s = g.s;
    hipStream_t st = s->stream;
    ZK_TRY(s->reserve(n_vars * sizeof(Fr) + 2 * scan_need(n + 8) + 65536));
    HFr bl[9], pin[5];
    memcpy(bl, blinders, sizeof bl);
    if (challenges) memcpy(pin, challenges, sizeof pin);
    const Fr* d_sol = (const Fr*)solution;
    if (!on_device) {
        Fr* t = (Fr*)s->alloc(n_vars * sizeof(Fr) + 16);
        if (n_vars) ZK_HIP(hipMemcpyAsync(t, solution, n_vars * sizeof(Fr), hipMemcpyHostToDevice, st));
        d_sol = t;
    }
    ScanBufs SB;
    ZK_TRY(scan_bufs(s, n + 8, &SB));
    Fr* d_vals = (Fr*)s->alloc(16 * sizeof(Fr));   // evaluation results
    // small-domain buffers (n + 8 elements each) inside the key's workspace
    const size_t S = n + 8;
    Fr* W = P->w_small;
    Fr *l_lag = W, *r_lag = W + S, *o_lag = W + 2 * S, *bl_ = W + 3 * S, *br_ = W + 4 * S, *bo_ = W + 5 * S, *bz_ = W + 6 * S, *num = W + 7 * S, *den = W + 8 * S,
       *qkc = W + 9 * S, *lin = W + 10 * S, *fh = W + 11 * S, *fold = W + 12 * S, *quo = W + 13 * S;
    Domain* d0;
    ZK_TRY(get_domain(s, st, logn, DOM_TW, &d0));

    // wall clock of the protocol's rounds (reported beside the kernels when profiling is on): each ends in a digest the next challenge needs
    auto lap_t = std::chrono::steady_clock::now();
    auto lap = [&](const char* name) {
        const auto t1 = std::chrono::steady_clock::now();
        prof_host(name, std::chrono::duration<double, std::milli>(t1 - lap_t).count());
        lap_t = t1;
    };
    // ---- l, r, o
    ZK_LAUNCH(s, st, "plonk_gather_lro", k_gather_lro, dim3(grid_of(n)), dim3(256), 0, d_sol, (const uint32_t*)P->xa, (const uint32_t*)P->xb, (const uint32_t*)P->xc,
              (uint32_t)npub, (uint32_t)P->n_constraints, (uint32_t)n, l_lag, r_lag, o_lag);
    std::vector<HFr> pub(npub);
    if (npub) ZK_HIP(hipMemcpyAsync(pub.data(), d_sol, npub * sizeof(Fr), hipMemcpyDeviceToHost, st));
    ZK_TRY(slot_sync(s, st));  // the public inputs are on the host (the transcript binds them); the stream is still almost empty here
    Fr* lag3[3] = {l_lag, r_lag, o_lag};
    Fr* can3[3] = {bl_, br_, bo_};
    Affine<HFp> c_lro[3], c_z, c_h[3], c_zopen, c_lin, c_batch;
    CommitChain chain;
    if (g_plonk_chain && !g_plonk_serial) ZK_TRY(chain.init(P, n + 3));
    // a group of independent commitments + the work the main stream does meanwhile
    hipStream_t commit_sync = st;  // the stream the polynomials of the next commit_group are produced on
    auto commit_group = [&](int cnt, const Fr* const* polys, const size_t* lens, Affine<HFp>* outs, const std::function<int()>& meanwhile) -> int {
        if (chain.ok) {
            for (int k = 0; k < cnt; k++) ZK_TRY(chain.start(k, st, polys[k], lens[k]));
            int rc = meanwhile();
            for (int k = 0; k < cnt; k++) {
                int r2 = chain.finish(k, &outs[k]);
                if (rc == ZK_OK) rc = r2;
            }
            return rc;
        }
        if (g_plonk_serial) {
            for (int k = 0; k < cnt; k++) ZK_TRY(commit(P, s, st, polys[k], lens[k], &outs[k]));
            return meanwhile();
        }
        if (cnt == 3 && g_plonk_batch3 && BatchCommit3::possible(P->srs, lens)) {
            BatchCommit3 bc;
            ZK_TRY(slot_sync(s, st));
            bc.start(P->srs, polys, lens[0]);
            int rc = meanwhile();
            const int r2 = bc.join();
            if (rc == ZK_OK) rc = r2;
            for (int k = 0; k < 3; k++) outs[k] = bc.out[k];
            return rc;
        }
        AsyncCommit ac[3];
        ZK_TRY(slot_sync(s, commit_sync));
        for (int k = 0; k < cnt; k++) ac[k].start(P, polys[k], lens[k]);
        int rc = meanwhile();
        for (int k = 0; k < cnt; k++) {
            int r2 = ac[k].join();
            if (rc == ZK_OK) rc = r2;
            outs[k] = ac[k].out;
        }
        return rc;
    };
    // canonical forms of l, r, o with their blinding (rounds 3-5 need them whichever way the digests are made)
    auto lro_canonical = [&]() -> int {
        for (int k = 0; k < 3; k++) {
            ZK_HIP(hipMemcpyAsync(can3[k], lag3[k], n * sizeof(Fr), hipMemcpyDeviceToDevice, st));
            ZK_HIP(hipMemsetAsync(can3[k] + n, 0, 8 * sizeof(Fr), st));
            ZK_TRY(to_canonical(s, st, can3[k], logn));
            BlindArgs B;
            B.k = 2;
            B.b[0] = to_dev(bl[2 * k]); B.b[1] = to_dev(bl[2 * k + 1]); B.b[2] = Fr::zero();
            ZK_LAUNCH(s, st, "plonk_blind", k_blind, dim3(1), dim3(64), 0, can3[k], (uint32_t)n, B);
        }
        return ZK_OK;
    };
    // the three commitments, while this stream already evaluates l, r, o on the big coset (no challenge needed for that).
    // Experiment ZKMI_PLONK_DEFER_LRO=1: those three transforms wait for round 2 (one commitment there instead of three here: round 1's three scalar
    // preparations then do not share the machine with them)
    static const bool defer_lro = ZK_EXP("ZKMI_PLONK_DEFER_LRO", 0) != 0;
    static const bool use_lagrange = ZK_EXP("ZKMI_PLONK_LAGRANGE", 1) != 0;  // 0: coefficients against the monomial SRS even when the key has its Lagrange form (A/B)
    const Fr* small5[5] = {bl_, br_, bo_, bz_, qkc};
    const size_t len5[5] = {n + 2, n + 2, n + 2, n + 3, n};
    const size_t lens_lag[3] = {n + 2, n + 2, n + 2};
    static const bool coset_side = ZK_EXP("ZKMI_PLONK_COSET_SIDE", 0) != 0;
    struct SideEvents {
        hipEvent_t e[2] = {nullptr, nullptr};
        hipStream_t side = nullptr;
        ~SideEvents() { if (side) (void)hipStreamSynchronize(side); for (auto& x : e) if (x) (void)hipEventDestroy(x); }
    } side_events;
    hipEvent_t* side_ev = side_events.e;
    bool side_pending = false, used_lagrange = false;
    if (P->lag_srs && use_lagrange && !chain.ok && !g_plonk_serial && BatchCommit3::possible(P->lag_srs, lens_lag)) {
        // From the WIRE VALUES against the SRS's Lagrange form (lagrange.hip): [l] = sum_i l_i [L_i(tau)] + b0 [tau^n - 1] + b1 [tau^(n+1) - tau] -- the same three
        // points, from scalars that are bits and words instead of uniform coefficients, and without waiting for the inverse transforms (they run meanwhile).
        for (int k = 0; k < 3; k++) {
            BlindArgs B;
            B.k = 2;
            B.b[0] = to_dev(bl[2 * k]); B.b[1] = to_dev(bl[2 * k + 1]); B.b[2] = Fr::zero();
            ZK_LAUNCH(s, st, "plonk_blind_tail", k_blind_tail, dim3(1), dim3(64), 0, lag3[k], (uint32_t)n, B);
        }
        ZK_TRY(slot_sync(s, st));
        used_lagrange = true;
        BatchCommit3 bc;
        bc.start(P->lag_srs, lag3, n + 2, true);
        int rc = lro_canonical();
        if (rc == ZK_OK && coset_side && !defer_lro) {  // experiment: the three 4n transforms on the slot's other stream, joined before the quotient kernel
            hipStream_t side = s->hi();
            side_events.side = side;
            if (hipEventCreateWithFlags(&side_ev[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&side_ev[1], hipEventDisableTiming) != hipSuccess ||
                hipEventRecord(side_ev[0], st) != hipSuccess || hipStreamWaitEvent(side, side_ev[0], 0) != hipSuccess)
                rc = set_err(ZK_ERR_HIP, "event setup failed");
            for (int k = 0; k < 3 && rc == ZK_OK; k++) rc = to_big_coset(s, side, P->w_big[k], small5[k], len5[k], P);
            if (rc == ZK_OK && hipEventRecord(side_ev[1], side) != hipSuccess) rc = set_err(ZK_ERR_HIP, "event record failed");
            side_pending = rc == ZK_OK;
        } else if (rc == ZK_OK && !defer_lro)
            for (int k = 0; k < 3 && rc == ZK_OK; k++) rc = to_big_coset(s, st, P->w_big[k], small5[k], len5[k], P);
        const int r2 = bc.join();
        if (rc == ZK_OK) rc = r2;
        ZK_TRY(rc);
        for (int k = 0; k < 3; k++) c_lro[k] = bc.out[k];
    } else {
        ZK_TRY(lro_canonical());
        const Fr* polys[3] = {bl_, br_, bo_};
        const size_t lens[3] = {n + 2, n + 2, n + 2};
        ZK_TRY(commit_group(3, polys, lens, c_lro, [&]() -> int {
            if (defer_lro) return ZK_OK;
            for (int k = 0; k < 3; k++) ZK_TRY(to_big_coset(s, st, P->w_big[k], small5[k], len5[k], P));
            return ZK_OK;
        }));
    }

    lap("plonk.round1_lro_committed");
    // ---- gamma, beta (transcript "gamma" binds the verifying key and the public inputs, then the three digests)
    FsTranscript fs{"gamma", "beta", "alpha", "zeta"};
    for (const Affine<HFp>* d : {&P->vk_s[0], &P->vk_s[1], &P->vk_s[2], &P->vk_ql, &P->vk_qr, &P->vk_qm, &P->vk_qo, &P->vk_qk}) fs.bind_g1(0, *d);
    for (const HFr& w : pub) fs.bind_fr(0, w);
    for (int k = 0; k < 3; k++) fs.bind_g1(0, c_lro[k]);
    HFr gamma = fs.challenge(0), beta = fs.challenge(1);
    if (challenges) { gamma = pin[0]; beta = pin[1]; }

    // ---- z
    const HFr u = P->u, uu = u * u;
    // experiment (ZKMI_PLONK_Z_SIDE=1, with the Lagrange-form round 1): round 1's coset transforms are still running on the main stream when the challenges arrive;
    // z's chain (short kernels) on the slot's high-priority stream instead of behind them, and its commitment starts as soon as THAT stream is done
    static const bool z_side = ZK_EXP("ZKMI_PLONK_Z_SIDE", 0) != 0;
    hipStream_t zst = (z_side && used_lagrange) ? s->hi() : st;
    ZK_LAUNCH(s, zst, "plonk_z_terms", k_z_terms, dim3(grid_of(n)), dim3(256), 0, (const Fr*)l_lag, (const Fr*)r_lag, (const Fr*)o_lag, (const Fr*)P->sig, (const Fr*)d0->tw,
              (uint32_t)n, to_dev(beta), to_dev(beta * u), to_dev(beta * uu), to_dev(gamma), num, den);
    {
        size_t lanes = (n + BINV_K - 1) / BINV_K;
        ZK_LAUNCH(s, zst, "plonk_batch_inverse", k_batch_inverse, dim3(grid_of(lanes)), dim3(256), 0, den, n, W + 14 * S);
    }
    ZK_LAUNCH(s, zst, "plonk_pscan_local", k_pscan_local, dim3(SB.nb), dim3(256), 0, (const Fr*)num, (const Fr*)den, n, SB.K, SB.t, SB.b);
    ZK_LAUNCH(s, zst, "plonk_pscan_blocks", k_pscan_blocks, dim3(1), dim3(1024), 0, SB.b, SB.nb);
    ZK_LAUNCH(s, zst, "plonk_pscan_apply", k_pscan_apply, dim3(SB.nb), dim3(256), 0, (const Fr*)num, (const Fr*)den, n, SB.K, (const Fr*)SB.t, (const Fr*)SB.b, bz_);
    ZK_HIP(hipMemsetAsync(bz_ + n, 0, 8 * sizeof(Fr), zst));
    ZK_TRY(to_canonical(s, zst, bz_, logn));
    {
        BlindArgs B;
        B.k = 3;
        for (int i = 0; i < 3; i++) B.b[i] = to_dev(bl[6 + i]);
        ZK_LAUNCH(s, zst, "plonk_blind", k_blind, dim3(1), dim3(64), 0, bz_, (uint32_t)n, B);
    }
    // commitment to z; meanwhile: qk completed with the public inputs (canonical), z and qk on the big coset
    commit_sync = zst;
    {
        const Fr* polys[1] = {bz_};
        const size_t lens[1] = {n + 3};
        ZK_TRY(commit_group(1, polys, lens, &c_z, [&]() -> int {
            if (defer_lro)
                for (int k = 0; k < 3; k++) ZK_TRY(to_big_coset(s, st, P->w_big[k], small5[k], len5[k], P));
            static const bool qk_ntt = ZK_EXP("ZKMI_PLONK_QK_NTT", 0) == 1;  // A/B switch: the literal sequence
            if (!qk_ntt && npub <= PLONK_PI_DIRECT_MAX && logN4 >= 2) {
                ZK_TRY(to_big_coset(s, st, P->w_big[3], small5[3], len5[3], P));
                ZK_LAUNCH(s, st, "plonk_qk_coset", k_qk_coset, dim3(grid_of(P->N4)), dim3(256), 0, (const Fr*)P->e_cqk, (const Fr*)P->e[7], (const Fr*)d_sol, (const Fr*)P->lqk,
                          (uint32_t)npub, logN4, P->log_rho, P->w_big[4]);
                return ZK_OK;
            }
            ZK_HIP(hipMemcpyAsync(qkc, P->lqk, n * sizeof(Fr), hipMemcpyDeviceToDevice, st));
            if (npub) ZK_HIP(hipMemcpyAsync(qkc, d_sol, npub * sizeof(Fr), hipMemcpyDeviceToDevice, st));
            ZK_TRY(to_canonical(s, st, qkc, logn));
            for (int k = 3; k < 5; k++) ZK_TRY(to_big_coset(s, st, P->w_big[k], small5[k], len5[k], P));
            return ZK_OK;
        }));
    }
    commit_sync = st;
    lap("plonk.round2_z_committed");
    fs.bind_g1(2, c_z);
    HFr alpha = fs.challenge(2);
    if (challenges) alpha = pin[2];

    // ---- quotient on the coset of the big domain
    {
        QuotArgs A;
        A.el = P->w_big[0]; A.er = P->w_big[1]; A.eo = P->w_big[2]; A.ez = P->w_big[3]; A.eqk = P->w_big[4];
        A.ql = P->e[0]; A.qr = P->e[1]; A.qm = P->e[2]; A.qo = P->e[3]; A.s1 = P->e[4]; A.s2 = P->e[5]; A.s3 = P->e[6]; A.l1 = P->e[7]; A.id = P->e[8];
        A.out = P->w_big[4];
        A.alpha = to_dev(alpha); A.beta = to_dev(beta); A.gamma = to_dev(gamma); A.beta_u = to_dev(beta * u); A.beta_uu = to_dev(beta * uu);
        A.logN4 = logN4; A.log_rho = P->log_rho;
        Domain* d1;
        ZK_TRY(get_domain(s, st, logN4, DOM_TW, &d1));
        // x = g * W^j  =>  x^n = g^n * (W^n)^j, W^n of order rho
        HFr gn = d1->coset, Wn = d1->gen;
        for (unsigned i = 0; i < logn; i++) { gn = gn.sqr(); Wn = Wn.sqr(); }
        HFr cur = gn;
        for (unsigned j = 0; j < (1u << P->log_rho); j++) {
            A.xn_inv[j] = to_dev((cur - HFr::one()).inv());
            cur = cur * Wn;
        }
        for (unsigned j = (1u << P->log_rho); j < 8; j++) A.xn_inv[j] = Fr::zero();
        if (side_pending) ZK_HIP(hipStreamWaitEvent(st, side_ev[1], 0));
        static const bool quot29 = ZK_EXP("ZKMI_PLONK_QUOT29", 1) != 0;  // A/B switch: 0 = the saturated 8 x 32-bit form
        if (quot29) ZK_LAUNCH(s, st, "plonk_quotient", k_quotient29, dim3(grid_of(N4)), dim3(256), 0, A);
        else ZK_LAUNCH(s, st, "plonk_quotient", k_quotient, dim3(grid_of(N4)), dim3(256), 0, A);
    }
    Fr* h = P->w_big[4];
    ZK_TRY(ntt_dev(s, st, h, logN4, 1, ZK_DIT, 1));  // LagrangeCoset (bit-reversed) -> canonical (regular)
    // the quotient has 3(n+2) coefficients iff the constraints hold: with a violated gate the remainder B = numerator mod (X^n - 1) is not
    // zero and shows up as B_j / (g^N4 - 1) in EVERY block of n coefficients above -- so all of h[3(n+2) .. N4) must vanish (upstream
    // fails earlier, in Solve)
    int* d_flag = (int*)s->alloc(64);
    int h_flag = 0;
    ZK_HIP(hipMemsetAsync(d_flag, 0, 4, st));
    ZK_LAUNCH(s, st, "plonk_quotient_check", k_any_nonzero, dim3(grid_of(N4 - 3 * (n + 2))), dim3(256), 0, (const Fr*)(h + 3 * (n + 2)), N4 - 3 * (n + 2), d_flag);
    ZK_HIP(hipMemcpyAsync(&h_flag, d_flag, 4, hipMemcpyDeviceToHost, st));
    {
        const Fr* polys[3] = {h, h + (n + 2), h + 2 * (n + 2)};
        const size_t lens[3] = {n + 2, n + 2, n + 2};
        ZK_TRY(commit_group(3, polys, lens, c_h, []() -> int { return ZK_OK; }));
        if (chain.ok) ZK_TRY(slot_sync(s, st));  // h_flag (chain mode does not synchronise this stream)
    }
    if (h_flag) return set_err(ZK_ERR_ARG, "the solution does not satisfy the constraint system (the quotient is not a polynomial)");
    lap("plonk.round3_quotient_committed");
    for (int k = 0; k < 3; k++) fs.bind_g1(3, c_h[k]);
    HFr zeta = fs.challenge(3);
    if (challenges) zeta = pin[3];

    // ---- evaluations at zeta, opening of z at omega * zeta
    const HFr zeta_sh = zeta * P->gen;
    const Fr* ev_p[5] = {bl_, br_, bo_, P->s1, P->s2};
    const size_t ev_len[5] = {n + 2, n + 2, n + 2, n, n};
    ZK_TRY(poly_eval_batch_dev(s, st, ev_p, ev_len, 5, zeta, SB, d_vals));
    ZK_TRY(poly_divide_dev(s, st, bz_, n + 3, zeta_sh, SB, quo, d_vals + 5));
    HFr ev[8];
    ZK_HIP(hipMemcpyAsync(ev, d_vals, 6 * sizeof(Fr), hipMemcpyDeviceToHost, st));
    ZK_TRY(slot_sync(s, st));  // ev[] is valid, the quotient of z is complete
    AsyncCommit azo;  // the opening of z: finished / joined at the end of the proof, nothing below depends on it
    if (chain.ok) ZK_TRY(chain.start(0, st, quo, n + 2));
    else if (g_plonk_serial) ZK_TRY(commit(P, s, st, quo, n + 2, &c_zopen));
    else azo.start(P, quo, n + 2);
    const HFr lz = ev[0], rz = ev[1], oz = ev[2], s1z = ev[3], s2z = ev[4], zu = ev[5];

    // ---- linearised polynomial
    HFr lin_cz, lin_cs3, lin_lag;
    {
        const HFr one = HFr::one();
        HFr c_s3 = (lz + beta * s1z + gamma) * (rz + beta * s2z + gamma) * zu * beta;
        HFr c_zz = HFr::zero() - (lz + beta * zeta + gamma) * (rz + beta * u * zeta + gamma) * (oz + beta * uu * zeta + gamma);
        HFr zn = zeta;
        for (unsigned i = 0; i < logn; i++) zn = zn.sqr();
        HFr lag = (zn - one) * (zeta - one).inv() * alpha * alpha * P->card_inv;
        lin_cz = c_zz; lin_cs3 = c_s3; lin_lag = lag;
        LinArgs A;
        A.bz = bz_; A.s3 = P->s3; A.ql = P->ql; A.qr = P->qr; A.qm = P->qm; A.qo = P->qo; A.cqk = P->cqk; A.out = lin;
        A.c_z = to_dev(c_zz); A.c_s3 = to_dev(c_s3); A.alpha = to_dev(alpha); A.rl = to_dev(lz * rz); A.lz = to_dev(lz); A.rz = to_dev(rz); A.oz = to_dev(oz);
        A.lag = to_dev(lag);
        A.n = (uint32_t)n; A.len = (uint32_t)(n + 3);
        ZK_LAUNCH(s, st, "plonk_linearized", k_linearized, dim3(grid_of(n + 3)), dim3(256), 0, A);
    }
    // Digest of the linearised polynomial.  It is a linear combination of polynomials whose commitments exist already -- the verifying key's and the proof's [Z] --
    // so commit(lin) = (alpha c_z + lag) [Z] + alpha c_s3 [S3] + lz rz [Qm] + lz [Ql] + rz [Qr] + oz [Qo] + [Qk]: the same group element as gnark's kzg.Commit of the
    // polynomial, for seven scalar multiplications on the host (while the GPU runs the kernels above) instead of an (n + 3)-point MSM.  Only for keys whose
    // digests are known to be consistent: Setup's, or -- for a key read from its wire image -- once a first proof has computed both and found them equal.
    // ZKMI_PLONK_LIN_MSM=1 forces the literal commitment (A/B switch).
    static const bool lin_msm = ZK_EXP("ZKMI_PLONK_LIN_MSM", 0) == 1;
    Affine<HFp> c_lin_by_linearity;
    {
        const HFr cz_tot = alpha * lin_cz + lin_lag, cs3_tot = alpha * lin_cs3;
        const Affine<HFp> pts[6] = {c_z, P->vk_s[2], P->vk_qm, P->vk_ql, P->vk_qr, P->vk_qo};
        const HFr ks[6] = {cz_tot, cs3_tot, lz * rz, lz, rz, oz};
        XYZZ<HFp> acc = host_multi_scalar_mul(pts, ks, 6);  // one shared doubling chain for the six (1.2 -> 0.45 ms: the host's share of round 4 is the round below 2^20 gates)
        acc.madd(P->vk_qk);
        c_lin_by_linearity = acc.to_affine();
    }
    const bool lin_direct = P->vk_consistent && !lin_msm;
    if (lin_direct) c_lin = c_lin_by_linearity;
    else if (chain.ok) ZK_TRY(chain.start(1, st, lin, n + 3));  // finished below, after the folded quotient and the last two evaluations are enqueued
    else ZK_TRY(commit(P, s, st, lin, n + 3, &c_lin));

    // ---- folded quotient h1 + zeta^(n+2) h2 + zeta^(2(n+2)) h3 and its digest
    HFr zp = HFr::one();
    {
        HFr base = zeta;
        for (size_t e = n + 2; e; e >>= 1) { if (e & 1) zp = zp * base; base = base.sqr(); }
    }
    {
        FoldArgs A = {};
        A.k = 3; A.n = (uint32_t)(n + 2); A.out = fh;
        for (int k = 0; k < 3; k++) { A.p[k] = h + k * (n + 2); A.len[k] = (uint32_t)(n + 2); }
        A.c[0] = Fr::one(); A.c[1] = to_dev(zp); A.c[2] = to_dev(zp * zp);
        ZK_LAUNCH(s, st, "plonk_fold", k_fold, dim3(grid_of(n + 2)), dim3(256), 0, A);
    }
    Affine<HFp> c_fh;
    {
        const Affine<HFp> pts[2] = {c_h[2], c_h[1]};
        const HFr ks[2] = {zp * zp, zp};
        XYZZ<HFp> fhd = host_multi_scalar_mul(pts, ks, 2);
        fhd.madd(c_h[0]);
        c_fh = fhd.to_affine();
    }

    // ---- kzg.BatchOpenSinglePoint of (foldedH, linPol, l, r, o, s1, s2) at zeta
    {
        const Fr* const p2[2] = {fh, lin};
        const size_t l2[2] = {n + 2, n + 3};
        ZK_TRY(poly_eval_batch_dev(s, st, p2, l2, 2, zeta, SB, d_vals + 6));
    }
    ZK_HIP(hipMemcpyAsync(ev + 6, d_vals + 6, 2 * sizeof(Fr), hipMemcpyDeviceToHost, st));
    ZK_TRY(slot_sync(s, st));
    lap("plonk.round4_evaluations_and_linearised");
    if (chain.ok && !lin_direct) ZK_TRY(chain.finish(1, &c_lin));
    if (!lin_direct && !lin_msm && c_lin.x == c_lin_by_linearity.x && c_lin.y == c_lin_by_linearity.y)
        const_cast<PlonkPK*>(P)->vk_consistent = true;  // (the key's workspace mutex is held for the whole proof) from the next proof on: by linearity
    const HFr claimed[7] = {ev[6], ev[7], lz, rz, oz, s1z, s2z};
    const Affine<HFp> digests[7] = {c_fh, c_lin, c_lro[0], c_lro[1], c_lro[2], P->vk_s[0], P->vk_s[1]};
    HFr kg;
    {
        FsTranscript ks{"gamma"};
        ks.bind_fr(0, zeta);
        for (const auto& d : digests) ks.bind_g1(0, d);
        for (const auto& v : claimed) ks.bind_fr(0, v);
        kg = ks.challenge(0);
        if (challenges) kg = pin[4];
    }
    {
        FoldArgs A = {};
        const Fr* pp[7] = {fh, lin, bl_, br_, bo_, P->s1, P->s2};
        const size_t pl[7] = {n + 2, n + 3, n + 2, n + 2, n + 2, n, n};
        A.k = 7; A.n = (uint32_t)(n + 3); A.out = fold;
        HFr acc = HFr::one();
        for (int k = 0; k < 7; k++) { A.p[k] = pp[k]; A.len[k] = (uint32_t)pl[k]; A.c[k] = to_dev(acc); acc = acc * kg; }
        ZK_LAUNCH(s, st, "plonk_fold", k_fold, dim3(grid_of(n + 3)), dim3(256), 0, A);
    }
    // dividePolyByXminusA(folded, foldedEvaluations, zeta): the recurrence yields the same quotient (folded(zeta) = sum gamma^i v_i)
    ZK_TRY(poly_divide_dev(s, st, fold, n + 3, zeta, SB, fold, d_vals + 8));
    if (chain.ok) {
        ZK_TRY(chain.start(2, st, fold, n + 2));
        ZK_TRY(chain.finish(0, &c_zopen));
        ZK_TRY(chain.finish(2, &c_batch));
    } else {
        ZK_TRY(commit(P, s, st, fold, n + 2, &c_batch));
        if (!g_plonk_serial) {
            ZK_TRY(azo.join());
            c_zopen = azo.out;
        }
    }
    lap("plonk.round5_openings_committed");
    // ---- Proof.WriteTo
    uint8_t* o = proof_out;
    for (const Affine<HFp>* d : {&c_lro[0], &c_lro[1], &c_lro[2], &c_z, &c_h[0], &c_h[1], &c_h[2]}) { g1_compress(*d, o); o += 32; }
    g1_compress(c_batch, o); o += 32;
    o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 7; o += 4;
    for (const HFr& v : claimed) { fr_to_be(v, o); o += 32; }
    g1_compress(c_zopen, o); o += 32;
    fr_to_be(zu, o); o += 32;
    return (o - proof_out) == ZK_PLONK_PROOF_BYTES ? ZK_OK : set_err(ZK_ERR_ARG, "internal: proof length");
}

}  // extern "C"
