// Radix-2 NTT over BN254 Fr on gfx950, with the exact in-place semantics of gnark-crypto v0.9.1
// `(*fft.Domain).FFT / FFTInverse` and `fft.BitReverse` (ecc/bn254/fr/fft; pinned at
// /root/reference/gnark_backend_ffi/go.mod:5; reached through groth16.Prove main.go:131 and plonk.Prove
// backend/plonk/plonk.go:67), plus gnark v0.8.0's `computeH` fused on the device.
//
// Structure: the log2(N) butterfly stages are grouped in passes; a pass keeps a tile of 2^k * L elements in LDS and
// runs k stages on it (one workgroup per tile), so HBM is touched once per pass instead of once per stage.
//   * tile  = every value of the k transformed index bits  x  L consecutive values of the lower bits  (so global
//     loads/stores are L*32-byte contiguous runs; the pass over the lowest bits has L = 1 and is fully contiguous)
//   * LDS layout = two uint4 arrays (low / high 16 bytes of each element): a wave reading consecutive elements
//     issues conflict-free ds_read_b128 / ds_write_b128
//   * the in-place DIF (natural -> bit-reversed) and DIT (bit-reversed -> natural) orders fall out of doing the
//     butterflies in place -- there is no transpose and no separate permutation pass
//   * coset / 1/N scalings are folded into the first stage's loads (pre table) or the last stage's stores (post)
// Roofline: 64 B per element per transform algorithmic (read + write once).  On CDNA4 the transform is VALU-bound, not HBM-bound: a butterfly is 327
// VALU instructions, 206 of them the 9 x 29-bit Montgomery product (162 v_mad_u64_u32, which issue at the rate of an add-with-carry -- 4.4-4.8 cycles per
// wave instruction, NOT quarter rate: DESIGN.md 3.1, tools/ubench.hip) -- see DESIGN.md 3.3.
#include <vector>

#include <algorithm>

#include "ctx.hpp"
#include "ff.hpp"
#include "host_ff.hpp"
#include "ntt.hpp"
#include "multidev.hpp"
#include "ff29.hpp"

namespace zkmi {

static constexpr unsigned TILE_LOG = 11;        // 2048 elements = 64 KiB of LDS per workgroup
static const unsigned K_STRIDED = (unsigned)std::min<long>(9, std::max<long>(2, ZK_EXP("ZKMI_NTT_KS", 9)));  // strided passes: k <= 9, rows of L >= 4 elements (128-byte runs); 2^20 = 11 + 9 bits = two passes
static constexpr unsigned NTT_THREADS = 256;

struct PowBasis {
    Fr pw[28];  // base^(2^b)
};

__device__ __forceinline__ unsigned bitrev_u32(unsigned i, unsigned logn) { return logn ? (__brev(i) >> (32 - logn)) : 0; }

// out[i] = scale * base^(i) or base^(bitrev(i))
__global__ void k_pow_table(Fr* out, size_t n, unsigned logn, PowBasis basis, Fr scale, int reversed, size_t offset) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned e = reversed ? bitrev_u32((unsigned)(i + offset), logn) : (unsigned)(i + offset);
    Fr acc = scale;
    for (unsigned b = 0; b < 28; b++)
        if ((e >> b) & 1) acc = acc * basis.pw[b];
    out[i] = acc;
}

struct PassArgs {
    Fr* data;
    const Fr* tw;        // w^i, i < N/2  (forward or inverse table)
    const Fr* pre;       // multiply element i by pre[i] on entry of the pass's first stage (or null)
    const Fr* post;      // multiply element i by post[i] on exit of the pass's last stage (or null)
    Fr post_const;       // if has_post_const: multiply by this constant on exit
    unsigned logn, bit_lo, k, logL;
    int dif, has_post_const;
    int canonical;       // 29-bit-limb passes: last pass of the transform -> canonical image; else a lazily reduced 256-bit intermediate
    const Fr* tw2;       // fused inverse-then-forward pass (k_ntt_pass29_if): twiddles of the forward half
    const Fr* src;       // 29-bit-limb passes: read the tile from here instead of `data` (first pass of an out-of-place transform) or null
    const Fr* sub;       // 29-bit-limb passes, last stage: after `post`, element i becomes (x - sub[i]) * post_const  (computeH's closing step) or null
    Fr* data2;           // 29-bit-limb passes launched with gridDim.y = 2 or 3: the workgroups with blockIdx.y = 1 / 2 run the same pass on these vectors
    const Fr* src2;      // (src2 -> data2, src3 -> data3).  In k_ntt_pass29_if the third vector only takes the inverse half, closed by post_const (computeH's c)
    Fr* data3;
    const Fr* src3;
    uint32_t unit_skip;  // 1: the stage on index bit 0 skips its product by the unit twiddle (0: A/B switch ZKMI_NTT_UNIT=0)
    uint32_t tw_and;     // EXPERIMENT (ZKMI_NTT_TWMASK): twiddle index mask -- 0xffffffff in production; a small mask makes every twiddle load an L1 hit
                         // (wrong results, right timing): the upper bound of what any twiddle-staging scheme could gain
};

__device__ __forceinline__ Fr lds_load(const uint4* lo, const uint4* hi, unsigned t) {
    uint4 a = lo[t], b = hi[t];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
__device__ __forceinline__ void lds_store(uint4* lo, uint4* hi, unsigned t, const Fr& v) {
    lo[t] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    hi[t] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
__device__ __forceinline__ Fr gload_fr(const Fr* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}

// G consecutive stages of the pass on one butterfly group per lane-iteration.  Stage index s counts from the start of the pass;
// DIF walks the tile bits downwards (q = k-1-s), DIT upwards (q = s).  ql = lowest tile bit of the group.
template <int G, bool DIF>
__device__ __forceinline__ void ntt_group(const PassArgs& A, uint4* lo, uint4* hi, size_t base, unsigned E, unsigned s0, bool first, bool last) {
    constexpr unsigned NE = 1u << G;
    const unsigned L = 1u << A.logL;
    const unsigned ql = DIF ? (A.k - s0 - G) : s0;
    for (unsigned u = threadIdx.x; u < (E >> G); u += NTT_THREADS) {
        const unsigned l = u & (L - 1), r = u >> A.logL;
        const unsigned mid0 = ((r >> ql) << (ql + G)) | (r & ((1u << ql) - 1));
        Fr x[NE];
#pragma unroll
        for (unsigned e = 0; e < NE; e++) {
            const unsigned t = ((mid0 | (e << ql)) << A.logL) + l;
            x[e] = lds_load(lo, hi, t);
            if (first && A.pre) x[e] = x[e] * gload_fr(A.pre + base + ((size_t)(mid0 | (e << ql)) << A.bit_lo) + l);
        }
#pragma unroll
        for (int sl = 0; sl < G; sl++) {
            const unsigned bitl = DIF ? (unsigned)(G - 1 - sl) : (unsigned)sl;  // bit inside the group (compile-time after unrolling)
            const unsigned b = A.bit_lo + ql + bitl;                               // global index bit
#pragma unroll
            for (unsigned e0 = 0; e0 < NE; e0++) {
                if (e0 & (1u << bitl)) continue;
                const unsigned e1 = e0 | (1u << bitl);
                const size_t g0 = base + ((size_t)(mid0 | (e0 << ql)) << A.bit_lo) + l;
                if (DIF) {
                    Fr sum = x[e0] + x[e1], dif = x[e0] - x[e1];
                    if (b != 0) {
                        size_t j = g0 & (((size_t)1 << b) - 1);
                        dif = dif * gload_fr(A.tw + (j << (A.logn - 1 - b)));
                    }
                    x[e0] = sum;
                    x[e1] = dif;
                } else {
                    Fr y = x[e1];
                    if (b != 0) {
                        size_t j = g0 & (((size_t)1 << b) - 1);
                        y = y * gload_fr(A.tw + (j << (A.logn - 1 - b)));
                    }
                    Fr sum = x[e0] + y, dif = x[e0] - y;
                    x[e0] = sum;
                    x[e1] = dif;
                }
            }
        }
#pragma unroll
        for (unsigned e = 0; e < NE; e++) {
            const unsigned t = ((mid0 | (e << ql)) << A.logL) + l;
            if (last) {
                if (A.post) x[e] = x[e] * gload_fr(A.post + base + ((size_t)(mid0 | (e << ql)) << A.bit_lo) + l);
                else if (A.has_post_const) x[e] = x[e] * A.post_const;
            }
            lds_store(lo, hi, t, x[e]);
        }
    }
}

__global__ __launch_bounds__(NTT_THREADS) void k_ntt_pass(PassArgs A) {
    prio_mid();
    extern __shared__ uint4 lds[];
    const unsigned E = 1u << (A.k + A.logL);
    uint4* lo = lds;
    uint4* hi = lds + E;
    const unsigned L = 1u << A.logL;
    const unsigned lo_blks = (1u << A.bit_lo) >> A.logL;  // >= 1
    const size_t tile = blockIdx.x;
    const size_t hi_idx = tile / lo_blks;
    const unsigned lo_blk = (unsigned)(tile % lo_blks);
    const size_t base = (hi_idx << (A.bit_lo + A.k)) + ((size_t)lo_blk << A.logL);
    uint4* g = reinterpret_cast<uint4*>(A.data);

    // ---- load tile: consecutive threads fetch consecutive 16-byte halves
    for (unsigned h = threadIdx.x; h < 2 * E; h += NTT_THREADS) {
        unsigned e = h >> 1, half = h & 1;
        unsigned mid = e >> A.logL, l = e & (L - 1);
        size_t gi = base + ((size_t)mid << A.bit_lo) + l;
        uint4 v = g[gi * 2 + half];
        (half ? hi : lo)[e] = v;
    }
    __syncthreads();

    // Stages are taken up to three at a time: a lane pulls the 2^G elements of one radix-2^G butterfly group out of LDS, runs
    // G stages on them in registers and puts them back -- one LDS round trip and one barrier per G stages instead of per stage.
    for (unsigned s0 = 0; s0 < A.k;) {
        const unsigned G = (A.k - s0 >= 3) ? 3 : (A.k - s0);
        const bool first = (s0 == 0), last = (s0 + G == A.k);
        if (A.dif) {
            if (G == 3) ntt_group<3, true>(A, lo, hi, base, E, s0, first, last);
            else if (G == 2) ntt_group<2, true>(A, lo, hi, base, E, s0, first, last);
            else ntt_group<1, true>(A, lo, hi, base, E, s0, first, last);
        } else {
            if (G == 3) ntt_group<3, false>(A, lo, hi, base, E, s0, first, last);
            else if (G == 2) ntt_group<2, false>(A, lo, hi, base, E, s0, first, last);
            else ntt_group<1, false>(A, lo, hi, base, E, s0, first, last);
        }
        s0 += G;
        __syncthreads();
    }

    for (unsigned h = threadIdx.x; h < 2 * E; h += NTT_THREADS) {
        unsigned e = h >> 1, half = h & 1;
        unsigned mid = e >> A.logL, l = e & (L - 1);
        size_t gi = base + ((size_t)mid << A.bit_lo) + l;
        g[gi * 2 + half] = (half ? hi : lo)[e];
    }
}

// ------------------------------------------------------------------------------------------------ 29-bit-limb passes
// Same pass structure with the butterflies in the unsaturated representation of ff29.hpp (Fr29): 1.4x fewer instructions per
// product, additions without carries.  Elements sit in LDS as 9 limbs (two 16-byte halves + the top limb) between the stage
// groups of a pass and as 8 packed words at its ends; A.tw is the w * 2^261 table.  Schedules and bounds: tools/u29_ntt_model.py.
__device__ __forceinline__ U29 lds_load9(const uint4* lo, const uint4* hi, const uint32_t* top, unsigned t) {
    uint4 a = lo[t], b = hi[t];
    U29 r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = top[t];
    return r;
}
__device__ __forceinline__ void lds_store9(uint4* lo, uint4* hi, uint32_t* top, unsigned t, const U29& v) {
    lo[t] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    hi[t] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    top[t] = v.l[8];
}

// What a pass does at its two ends is a compile-time MODE, resolved against the kernel arguments where it is used (a struct of pointers built once per kernel
// keeps them all in scalar registers for the whole kernel, and the hot loop pays for the spills: measured 4 % on a 2^24-point transform):
//   0  a plain pass: A.tw, entry table A.pre, exit table A.post or the constant A.post_const, canonical image if A.canonical
//   1  inverse half of k_ntt_pass29_if: A.tw, exit table A.post, limbs stay unpacked
//   2  forward half of k_ntt_pass29_if: A.tw2, no tables, canonical image if A.canonical
//   3  k_ntt_pass29_if's third vector, a plain FFTInverse ending: A.tw, * A.post_const, canonical image
// SUB (compile-time too: its extra live values would cost every other pass a wave of occupancy) adds computeH's closing step.
template <int G, bool DIF, unsigned THREADS, bool SUB, int MODE, bool UNIT>
__device__ __forceinline__ void ntt_group29(const PassArgs& A, uint4* lo, uint4* hi, uint32_t* top, size_t base, unsigned E, unsigned s0,
                                            bool load_packed, bool apply_pre, bool apply_post, bool store_packed) {
    constexpr unsigned NE = 1u << G;
    const unsigned L = 1u << A.logL;
    const unsigned ql = DIF ? (A.k - s0 - G) : s0;
    for (unsigned u = threadIdx.x; u < (E >> G); u += THREADS) {
        const unsigned l = u & (L - 1), r = u >> A.logL;
        const unsigned mid0 = ((r >> ql) << (ql + G)) | (r & ((1u << ql) - 1));
        U29 x[NE];
#pragma unroll
        for (unsigned e = 0; e < NE; e++) {
            const unsigned t = ((mid0 | (e << ql)) << A.logL) + l;
            x[e] = load_packed ? u29_unpack(lds_load(lo, hi, t)) : lds_load9(lo, hi, top, t);
            if (MODE == 0 && apply_pre && A.pre) x[e] = u29r_mul(x[e], u29r_load5(gload_fr(A.pre + base + ((size_t)(mid0 | (e << ql)) << A.bit_lo) + l)));
        }
#pragma unroll
        for (int sl = 0; sl < G; sl++) {
            const unsigned bitl = DIF ? (unsigned)(G - 1 - sl) : (unsigned)sl;
            const unsigned b = A.bit_lo + ql + bitl;
#pragma unroll
            for (unsigned e0 = 0; e0 < NE; e0++) {
                if (e0 & (1u << bitl)) continue;
                const unsigned e1 = e0 | (1u << bitl);
                const size_t g0 = base + ((size_t)(mid0 | (e0 << ql)) << A.bit_lo) + l;
                const size_t j = g0 & (((size_t)1 << b) - 1);
                // the stage on index bit 0 (the last of a DIF transform, the first of a DIT one) has no twiddles: its product by the unit is replaced by a
                // partial reduction (DIF) or dropped (DIT) -- one product in 20 at 2^20; same values mod r, bounds in tools/u29_ntt_model.py (`unit`).
                // UNIT is a property of the GROUP (it holds index bit 0), chosen by the caller: a run-time test here would sit in every group's bitl == 0
                // stage and cut the butterflies' instruction streams apart -- measured: 3 % on a 2^26-point transform, more than the skipped product gives
                // In that group the stage on bit 1 has unit twiddles too, for the butterflies whose low element has bit 0 clear (e0 even): half of them.
                // (Stage 2's quarter would meet sums of four elements, whose limbs no bias covers.)
                constexpr bool unit_ct = UNIT;
                const bool unit = unit_ct && (bitl == 0 || (bitl == 1 && (e0 & 1u) == 0));
                U29 w;
                if (!unit) w = u29_unpack(gload_fr((MODE == 2 ? A.tw2 : A.tw) + ((j << (A.logn - 1 - b)) & A.tw_and)));
                if (DIF) {
                    U29 d;
                    if (sl == 0) d = u29r_sub<16>(x[e0], x[e1]);
                    else if (sl == 1) d = u29r_sub<24>(x[e0], x[e1]);
                    else d = u29r_sub<40>(x[e0], x[e1]);
                    x[e0] = u29_wnorm(u29_add(x[e0], x[e1]));
                    x[e1] = unit ? u29r_reduce(u29_wnorm(d)) : u29r_mul(d, w);
                } else {
                    const U29 t = unit ? x[e1] : u29r_mul(x[e1], w);
                    // a unit twiddle on bit 1 meets an element that is already the sum of two (< 4.4 r, limbs < 2^30): 16 r instead of 4 r
                    x[e1] = u29_wnorm((unit && bitl == 1) ? u29r_sub<16>(x[e0], t) : u29r_sub<4>(x[e0], t));
                    x[e0] = u29_add(x[e0], t);
                }
            }
        }
        if (DIF) {
            x[0] = u29r_reduce(x[0]);  // the all-sums output is the only one that grows (8x per group)
        } else {
#pragma unroll
            for (unsigned e = 0; e < NE; e++) x[e] = u29_wnorm(x[e]);
        }
#pragma unroll
        for (unsigned e = 0; e < NE; e++) {
            const unsigned t = ((mid0 | (e << ql)) << A.logL) + l;
            if (apply_post) {
                if ((MODE == 0 || MODE == 1) && A.post) x[e] = u29r_mul(x[e], u29r_load5(gload_fr(A.post + base + ((size_t)(mid0 | (e << ql)) << A.bit_lo) + l)));
                else if (MODE == 3 || (MODE == 0 && A.has_post_const)) x[e] = u29r_mul(x[e], u29r_load5(A.post_const));
                if (SUB) {  // x < 2 r after the product above; sub[i] canonical: the difference stays below 6 r, a legal multiplicand (tools/u29_ntt_model.py)
                    const U29 c = u29_unpack(gload_fr(A.sub + base + ((size_t)(mid0 | (e << ql)) << A.bit_lo) + l));
                    x[e] = u29r_mul(u29r_sub<4>(x[e], c), u29r_load5(A.post_const));
                }
            }
            if (store_packed) lds_store(lo, hi, t, u29r_pack(u29r_reduce(x[e]), MODE == 3 || A.canonical != 0));
            else lds_store9(lo, hi, top, t, x[e]);
        }
    }
}

// all k stages of a pass, GMAX at a time.  packed_in / packed_out: the tile sits in LDS as 8 packed words before / after.
template <int GMAX, unsigned THREADS, bool DIF, bool SUB, int MODE>
__device__ __forceinline__ void ntt_stages29(const PassArgs& A, uint4* lo, uint4* hi, uint32_t* top, size_t base, unsigned E, bool packed_in, bool packed_out) {
    for (unsigned s0 = 0; s0 < A.k;) {
        const unsigned G = (A.k - s0 >= (unsigned)GMAX) ? (unsigned)GMAX : (A.k - s0);
        const bool first = (s0 == 0), last = (s0 + G == A.k);
        const bool lp = first && packed_in, sp = last && packed_out;
        const bool ug = A.unit_skip && A.bit_lo == 0 && (DIF ? last : first);  // the group that holds index bit 0 (uniform)
        if (ug) {
            if (GMAX >= 3 && G == 3) ntt_group29<(GMAX >= 3 ? 3 : 1), DIF, THREADS, SUB, MODE, true>(A, lo, hi, top, base, E, s0, lp, first, last, sp);
            else if (G == 2) ntt_group29<2, DIF, THREADS, SUB, MODE, true>(A, lo, hi, top, base, E, s0, lp, first, last, sp);
            else ntt_group29<1, DIF, THREADS, SUB, MODE, true>(A, lo, hi, top, base, E, s0, lp, first, last, sp);
        } else {
            if (GMAX >= 3 && G == 3) ntt_group29<(GMAX >= 3 ? 3 : 1), DIF, THREADS, SUB, MODE, false>(A, lo, hi, top, base, E, s0, lp, first, last, sp);
            else if (G == 2) ntt_group29<2, DIF, THREADS, SUB, MODE, false>(A, lo, hi, top, base, E, s0, lp, first, last, sp);
            else ntt_group29<1, DIF, THREADS, SUB, MODE, false>(A, lo, hi, top, base, E, s0, lp, first, last, sp);
        }
        s0 += G;
        __syncthreads();
    }
}

// which vector a workgroup works on (gridDim.y = 1 .. 3): resolved once, in scalar registers.  Masks instead of ?: on purpose -- a conditional between two
// members of the kernel-argument struct is a conditional between their ADDRESSES, which makes the compiler keep a copy of the whole struct in scratch memory
__device__ __forceinline__ uintptr_t pick3(uintptr_t v0, uintptr_t v1, uintptr_t v2) {
    const uintptr_t m1 = (uintptr_t)0 - (uintptr_t)(blockIdx.y == 1), m2 = (uintptr_t)0 - (uintptr_t)(blockIdx.y == 2);
    return (v0 & ~(m1 | m2)) | (v1 & m1) | (v2 & m2);
}
__device__ __forceinline__ Fr* pass_data(const PassArgs& A) { return reinterpret_cast<Fr*>(pick3((uintptr_t)A.data, (uintptr_t)A.data2, (uintptr_t)A.data3)); }
__device__ __forceinline__ const Fr* pass_src(const PassArgs& A, const Fr* data) {
    const uintptr_t src = pick3((uintptr_t)A.src, (uintptr_t)A.src2, (uintptr_t)A.src3);
    return src ? reinterpret_cast<const Fr*>(src) : data;
}
template <unsigned THREADS>
__device__ __forceinline__ void tile_copy_in(const Fr* src, unsigned logL, unsigned bit_lo, uint4* lo, uint4* hi, size_t base, unsigned E) {
    const unsigned L = 1u << logL;
    // the pointer went through integer masks (pick3): tell the compiler again that it is global memory, or it emits flat_load / flat_store
#ifdef __HIP_DEVICE_COMPILE__
    typedef __attribute__((address_space(1))) const uint4 g_uint4;
    const g_uint4* g = (const g_uint4*)(uintptr_t)src;
#else
    const uint4* g = reinterpret_cast<const uint4*>(src);  // host pass: never executed
#endif
    for (unsigned h = threadIdx.x; h < 2 * E; h += THREADS) {
        unsigned e = h >> 1, half = h & 1;
        unsigned mid = e >> logL, l = e & (L - 1);
        size_t gi = base + ((size_t)mid << bit_lo) + l;
        (half ? hi : lo)[e] = g[gi * 2 + half];
    }
}
template <unsigned THREADS>
__device__ __forceinline__ void tile_copy_out(Fr* dst, unsigned logL, unsigned bit_lo, const uint4* lo, const uint4* hi, size_t base, unsigned E) {
    const unsigned L = 1u << logL;
#ifdef __HIP_DEVICE_COMPILE__
    typedef __attribute__((address_space(1))) uint4 g_uint4;
    g_uint4* g = (g_uint4*)(uintptr_t)dst;
#else
    uint4* g = reinterpret_cast<uint4*>(dst);
#endif
    for (unsigned h = threadIdx.x; h < 2 * E; h += THREADS) {
        unsigned e = h >> 1, half = h & 1;
        unsigned mid = e >> logL, l = e & (L - 1);
        size_t gi = base + ((size_t)mid << bit_lo) + l;
        g[gi * 2 + half] = (half ? hi : lo)[e];
    }
}

// 512 lanes: two workgroups per CU (4 waves per SIMD, <= 128 VGPRs); 256 lanes (radix-8 groups, A/B variant): two waves per SIMD.  The register counts are
// what -Rpass-analysis=kernel-resource-usage shows (123 / 119 for the 512-lane kernels); asking for them with __launch_bounds__(512, 4) gives the same
// counts and a schedule that is 1.5 % slower on a 2^26-point transform (measured), so the bound stays implicit -- check the remark when touching the kernels.
#define ZK_NTT_BOUNDS(THREADS) __launch_bounds__(THREADS)

template <int GMAX, unsigned THREADS, bool SUB = false>
__global__ ZK_NTT_BOUNDS(THREADS) void k_ntt_pass29(PassArgs A) {
    prio_mid();
    extern __shared__ uint4 lds[];
    const unsigned E = 1u << (A.k + A.logL);
    uint4* lo = lds;
    uint4* hi = lds + E;
    uint32_t* top = reinterpret_cast<uint32_t*>(lds + 2 * E);
    const unsigned lo_blks = (1u << A.bit_lo) >> A.logL;  // >= 1
    const size_t tile = blockIdx.x;
    const size_t hi_idx = tile / lo_blks;
    const unsigned lo_blk = (unsigned)(tile % lo_blks);
    const size_t base = (hi_idx << (A.bit_lo + A.k)) + ((size_t)lo_blk << A.logL);
    Fr* const data = pass_data(A);
    tile_copy_in<THREADS>(pass_src(A, data), A.logL, A.bit_lo, lo, hi, base, E);
    __syncthreads();
    if (A.dif) ntt_stages29<GMAX, THREADS, true, SUB, 0>(A, lo, hi, top, base, E, true, true);
    else ntt_stages29<GMAX, THREADS, false, SUB, 0>(A, lo, hi, top, base, E, true, true);
    tile_copy_out<THREADS>(data, A.logL, A.bit_lo, lo, hi, base, E);
}

// computeH runs FFTInverse(DIF) immediately followed by FFT(DIT, coset) on the same vector: the inverse transform ENDS with the
// contiguous pass over the low index bits and the forward transform BEGINS with it, on the same tiles -- so the two passes are one
// kernel: load tile, k DIF stages (A.tw), * A.post (1/N * g^bitrev(i)), k DIT stages (A.tw2), store.  One HBM round trip saved
// per vector.
template <int GMAX, unsigned THREADS>
__global__ ZK_NTT_BOUNDS(THREADS) void k_ntt_pass29_if(PassArgs A) {
    prio_mid();
    extern __shared__ uint4 lds[];
    const unsigned E = 1u << (A.k + A.logL);
    uint4* lo = lds;
    uint4* hi = lds + E;
    uint32_t* top = reinterpret_cast<uint32_t*>(lds + 2 * E);
    const size_t base = (size_t)blockIdx.x << A.k;  // contiguous tiles only (bit_lo = 0, logL = 0)
    Fr* const data = pass_data(A);
    tile_copy_in<THREADS>(pass_src(A, data), 0, 0, lo, hi, base, E);
    __syncthreads();
    if (blockIdx.y == 2) {  // the third vector: a plain FFTInverse ending (* post_const, canonical image)
        ntt_stages29<GMAX, THREADS, true, false, 3>(A, lo, hi, top, base, E, true, true);
    } else {
        ntt_stages29<GMAX, THREADS, true, false, 1>(A, lo, hi, top, base, E, true, false);   // inverse half; its post table is applied, limbs stay unpacked
        ntt_stages29<GMAX, THREADS, false, false, 2>(A, lo, hi, top, base, E, false, true);  // forward half
    }
    tile_copy_out<THREADS>(data, 0, 0, lo, hi, base, E);
}

// a[i] *= t[i]  (used when a transform has no stage to fold a scaling into: N == 1)
__global__ void k_scale_table(Fr* a, const Fr* t, size_t n) {
    prio_mid();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = a[i] * gload_fr(t + i);
}

__global__ void k_bit_reverse(Fr* a, unsigned logn) {
    prio_mid();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)1 << logn;
    if (i >= n) return;
    size_t j = bitrev_u32((unsigned)i, logn);
    if (i < j) {
        Fr x = gload_fr(a + i), y = gload_fr(a + j);
        a[i] = y;
        a[j] = x;
    }
}

// h = (a*b - c) * den    (gnark computeH pointwise step)
__global__ void k_h_pointwise(Fr* a, const Fr* b, const Fr* c, Fr den, size_t n) {
    prio_mid();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr x = gload_fr(a + i) * gload_fr(b + i) - gload_fr(c + i);
    a[i] = x * den;
}

// h = (u - c) * den: the last step of computeH when c stays in coefficient form (see compute_h_inplace)
__global__ void k_h_final(Fr* u, const Fr* c, Fr den, size_t n) {
    prio_mid();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u[i] = (gload_fr(u + i) - gload_fr(c + i)) * den;
}

__global__ void k_fr_mul(Fr* out, const Fr* a, const Fr* b, size_t n) {
    prio_mid();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gload_fr(a + i) * gload_fr(b + i);
}

// ---------------------------------------------------------------------------------------------- domain tables
static const uint64_t ROOT_2_28_MONT[4] = {0x636e735580d13d9cULL, 0xa22bf3742445ffd6ULL, 0x56452ac01eb203d8ULL, 0x1860ef942963f9e7ULL};

static std::mutex g_dom_mu;
static std::map<unsigned, Domain*> g_domains;  // by (device entry, log n): the tables live in the HBM of the entry that built them

static Fr to_dev(const HFr& h) {
    Fr r;
    memcpy(&r, &h, 32);
    return r;
}

static int make_pow_table(Slot* s, hipStream_t st, Fr** out, size_t n, unsigned logn, HFr base, HFr scale, int reversed, size_t offset = 0) {
    ZK_HIP(hipMalloc((void**)out, (n ? n : 1) * sizeof(Fr)));
    PowBasis pb;
    HFr p = base;
    for (int b = 0; b < 28; b++) {
        pb.pw[b] = to_dev(p);
        p = p.sqr();
    }
    unsigned grid = (unsigned)((n + 255) / 256);
    ZK_LAUNCH(s, st, "pow_table", k_pow_table, dim3(grid ? grid : 1), dim3(256), 0, *out, n, logn, pb, to_dev(scale), reversed, offset);
    return ZK_OK;
}

int get_domain(Slot* s, hipStream_t st, unsigned logn, unsigned need, Domain** out) {
    if (logn > 28) return set_err(ZK_ERR_ARG, "log_n = %u exceeds Fr two-adicity 28", logn);
    std::lock_guard<std::mutex> lk(g_dom_mu);
    Domain*& d = g_domains[((unsigned)current_entry() << 8) | logn];
    if (!d) {
        d = new Domain();
        d->logn = logn;
        HFr w;
        memcpy(&w, ROOT_2_28_MONT, 32);
        for (unsigned i = logn; i < 28; i++) w = w.sqr();
        d->gen = w;
        d->gen_inv = w.inv();
        HFr n = HFr{{(uint64_t)1 << logn, 0, 0, 0}}.to_mont();
        d->card_inv = n.inv();
        d->coset = HFr{{5, 0, 0, 0}}.to_mont();
        d->coset_inv = d->coset.inv();
    }
    size_t N = (size_t)1 << logn, H = N > 1 ? N / 2 : 1;
    bool made = false;
    if ((need & DOM_TW) && !d->tw) { ZK_TRY(make_pow_table(s, st, &d->tw, H, logn, d->gen, HFr::one(), 0)); made = true; }
    if ((need & DOM_TW_INV) && !d->tw_inv) { ZK_TRY(make_pow_table(s, st, &d->tw_inv, H, logn, d->gen_inv, HFr::one(), 0)); made = true; }
    // the same powers times 2^5: w * 2^261 mod r, the multiplier form of the 29-bit-limb passes
    const HFr two5 = HFr{{32, 0, 0, 0}}.to_mont();
    if ((need & DOM_TW) && !d->tw29) { ZK_TRY(make_pow_table(s, st, &d->tw29, H, logn, d->gen, two5, 0)); made = true; }
    if ((need & DOM_TW_INV) && !d->tw29_inv) { ZK_TRY(make_pow_table(s, st, &d->tw29_inv, H, logn, d->gen_inv, two5, 0)); made = true; }
    if ((need & DOM_COSET) && !d->coset_tab) { ZK_TRY(make_pow_table(s, st, &d->coset_tab, N, logn, d->coset, HFr::one(), 0)); made = true; }
    if ((need & DOM_COSET_REV) && !d->coset_rev) { ZK_TRY(make_pow_table(s, st, &d->coset_rev, N, logn, d->coset, HFr::one(), 1)); made = true; }
    if ((need & DOM_COSET_INV_N) && !d->coset_inv_n) { ZK_TRY(make_pow_table(s, st, &d->coset_inv_n, N, logn, d->coset_inv, d->card_inv, 0)); made = true; }
    if ((need & DOM_COSET_INV_N_REV) && !d->coset_inv_n_rev) { ZK_TRY(make_pow_table(s, st, &d->coset_inv_n_rev, N, logn, d->coset_inv, d->card_inv, 1)); made = true; }
    if ((need & DOM_COSET_REV_N) && !d->coset_rev_n) { ZK_TRY(make_pow_table(s, st, &d->coset_rev_n, N, logn, d->coset, d->card_inv, 1)); made = true; }
    if (made) ZK_HIP(hipStreamSynchronize(st));  // tables are shared by every stream from here on
    *out = d;
    return ZK_OK;
}

// Runs the log2(N) stages of one transform as a sequence of tile passes.
// radix-4 stage groups with 512 lanes per tile (4 waves per SIMD) instead of radix-8 with 256 (2 waves per SIMD): measured 8 % faster --
// the waves of a pass spend ~40 % of their time parked on twiddle / tile loads and barriers, which more resident waves hide.
// ZKMI_NTT_G2=0 selects the radix-8 variant (A/B switch).
static const bool g_ntt_g2 = (ZK_EXP("ZKMI_NTT_G2", 1) != 0);
static const bool g_ntt_saturated = ZK_EXP("ZKMI_NTT_SAT", 0) == 1;  // A/B switch: the 8 x 32-bit butterflies

struct PassPlan { unsigned bit_lo, k, logL; };
// split of the index bits: the lowest kc bits form the contiguous pass (L = 1); the rest go to strided passes (increasing bit order)
static std::vector<PassPlan> plan_passes(unsigned logn) {
    unsigned kc = logn < TILE_LOG ? logn : TILE_LOG;
    unsigned rest = logn - kc;
    unsigned np = (rest + K_STRIDED - 1) / K_STRIDED;
    std::vector<PassPlan> passes;
    passes.push_back({0, kc, 0});
    unsigned bit = kc;
    for (unsigned i = 0; i < np; i++) {
        unsigned k = (rest - (bit - kc) + (np - i) - 1) / (np - i);
        unsigned logL = TILE_LOG - k;
        if (logL > bit) logL = bit;  // rows of L consecutive low-order neighbours; the tile always holds up to 2^TILE_LOG elements
        passes.push_back({bit, k, logL});
        bit += k;
    }
    return passes;
}

static const uint32_t g_tw_and = (uint32_t)ZK_EXP("ZKMI_NTT_TWMASK", 0xffffffffL);
static const uint32_t g_unit_skip = ZK_EXP("ZKMI_NTT_UNIT", 1) != 0;
static int launch_pass(Slot* s, hipStream_t st, const PassArgs& A_, bool sat) {
    PassArgs A = A_;
    A.tw_and = g_tw_and;
    A.unit_skip = g_unit_skip;
    unsigned E = 1u << (A.k + A.logL);
    size_t tiles = ((size_t)1 << A.logn) / E;
    const char* name = A.logL ? "ntt_pass_strided" : "ntt_pass_contig";
    const unsigned ny = sat ? 1 : 1 + (A.data2 ? 1 : 0) + (A.data2 && A.data3 ? 1 : 0);
    if (sat) ZK_LAUNCH(s, st, name, k_ntt_pass, dim3((unsigned)tiles), dim3(NTT_THREADS), (size_t)E * 32, A);
    else if (A.sub && g_ntt_g2) ZK_LAUNCH(s, st, name, (k_ntt_pass29<2, 512, true>), dim3((unsigned)tiles, ny), dim3(512), (size_t)E * 36, A);
    else if (A.sub) ZK_LAUNCH(s, st, name, (k_ntt_pass29<3, 256, true>), dim3((unsigned)tiles, ny), dim3(NTT_THREADS), (size_t)E * 36, A);
    else if (g_ntt_g2) ZK_LAUNCH(s, st, name, (k_ntt_pass29<2, 512>), dim3((unsigned)tiles, ny), dim3(512), (size_t)E * 36, A);
    else ZK_LAUNCH(s, st, name, (k_ntt_pass29<3, 256>), dim3((unsigned)tiles, ny), dim3(NTT_THREADS), (size_t)E * 36, A);
    return ZK_OK;
}

// Runs the log2(N) stages of one transform as a sequence of tile passes.
static int run_passes(Slot* s, hipStream_t st, Fr* data, const Domain* dom, int inverse, int dif, const Fr* pre, const Fr* post,
                      const Fr* post_const, const Fr* src = nullptr, const Fr* sub = nullptr) {
    // sub (29-bit-limb passes only, with `post` AND `post_const`): the last stage leaves (x * post[i] - sub[i]) * post_const
    const unsigned logn = dom->logn;
    const bool sat = g_ntt_saturated;
    const Fr* tw = sat ? (inverse ? dom->tw_inv : dom->tw) : (inverse ? dom->tw29_inv : dom->tw29);
    if (logn == 0) {
        if (src && src != data) ZK_HIP(hipMemcpyAsync(data, src, sizeof(Fr), hipMemcpyDeviceToDevice, st));
        if (pre) ZK_LAUNCH(s, st, "ntt_scale", k_scale_table, dim3(1), dim3(64), 0, data, pre, (size_t)1);
        if (post) ZK_LAUNCH(s, st, "ntt_scale", k_scale_table, dim3(1), dim3(64), 0, data, post, (size_t)1);
        return ZK_OK;  // 1/N = 1
    }
    if (sat && src && src != data) {  // the saturated kernels (A/B switch) work in place only
        ZK_HIP(hipMemcpyAsync(data, src, sizeof(Fr) << logn, hipMemcpyDeviceToDevice, st));
        src = nullptr;
    }
    std::vector<PassPlan> passes = plan_passes(logn);
    size_t npass = passes.size();
    for (size_t idx = 0; idx < npass; idx++) {
        // DIF walks the bits from the top, DIT from the bottom
        const PassPlan& p = dif ? passes[npass - 1 - idx] : passes[idx];
        PassArgs A;
        A.data2 = nullptr; A.src2 = nullptr; A.data3 = nullptr; A.src3 = nullptr;
        A.data = data; A.tw = tw; A.tw2 = nullptr; A.src = (idx == 0 && src && src != data) ? src : nullptr; A.logn = logn; A.bit_lo = p.bit_lo; A.k = p.k; A.logL = p.logL; A.dif = dif;
        A.pre = (idx == 0) ? pre : nullptr;
        A.post = (idx + 1 == npass) ? post : nullptr;
        A.has_post_const = (idx + 1 == npass && post_const && !post) ? 1 : 0;
        A.sub = (idx + 1 == npass && !sat) ? sub : nullptr;
        if (A.has_post_const || A.sub) A.post_const = *post_const; else A.post_const = Fr::zero();
        A.canonical = (idx + 1 == npass) ? 1 : 0;
        ZK_TRY(launch_pass(s, st, A, sat));
    }
    return ZK_OK;
}

// FFTInverse(DIF) with `mid` applied at its end (1/N and whatever scaling follows), then FFT(DIT): as run_passes twice, but the two
// contiguous passes in the middle are ONE kernel (k_ntt_pass29_if).
static const bool g_ntt_fuse_if = (ZK_EXP("ZKMI_NTT_FUSE", 1) != 0);  // A/B switch
static int run_inverse_forward(Slot* s, hipStream_t st, Fr* data, const Domain* dom, const Fr* mid, const Fr* src = nullptr, Fr* data2 = nullptr,
                               const Fr* src2 = nullptr, Fr* data3 = nullptr, const Fr* src3 = nullptr, const Fr* end3 = nullptr) {
    // src (optional): the input lives there and stays untouched -- the first pass reads it and writes `data`
    // data2 / src2 (optional): a second vector taken through the same passes by the same launches (gridDim.y = 2)
    // data3 / src3 / end3 (optional, with data2): a third vector that only takes the INVERSE transform, closed by the constant *end3 (computeH's c), in the
    // launches of the inverse half (gridDim.y = 3)
    const unsigned logn = dom->logn;
    if (g_ntt_saturated || !g_ntt_fuse_if || logn == 0) {
        if (data2) {  // the unfused A/B variants take the vectors one after the other
            ZK_TRY(run_inverse_forward(s, st, data, dom, mid, src));
            ZK_TRY(run_inverse_forward(s, st, data2, dom, mid, src2));
            return data3 ? run_passes(s, st, data3, dom, 1, 1, nullptr, nullptr, end3, src3) : ZK_OK;
        }
        if (src && src != data) ZK_HIP(hipMemcpyAsync(data, src, sizeof(Fr) << logn, hipMemcpyDeviceToDevice, st));
        ZK_TRY(run_passes(s, st, data, dom, 1, 1, nullptr, mid, nullptr));
        return run_passes(s, st, data, dom, 0, 0, nullptr, nullptr, nullptr);
    }
    std::vector<PassPlan> passes = plan_passes(logn);
    const size_t npass = passes.size();
    PassArgs A;
    A.data = data; A.logn = logn; A.pre = nullptr; A.post = nullptr; A.has_post_const = 0; A.post_const = Fr::zero(); A.tw2 = nullptr;
    A.sub = nullptr;
    A.tw_and = g_tw_and;
    A.unit_skip = g_unit_skip;
    A.src = (src && src != data) ? src : nullptr;  // consumed by whichever pass runs first
    A.data2 = data2;
    A.src2 = (data2 && src2 && src2 != data2) ? src2 : nullptr;
    A.data3 = data2 ? data3 : nullptr;
    A.src3 = (A.data3 && src3 && src3 != data3) ? src3 : nullptr;
    if (A.data3) A.post_const = *end3;
    const unsigned ny = 1 + (data2 ? 1 : 0) + (A.data3 ? 1 : 0);
    for (size_t idx = npass - 1; idx >= 1; idx--) {  // strided passes of the inverse transform, top bits first
        const PassPlan& p = passes[idx];
        A.tw = dom->tw29_inv; A.bit_lo = p.bit_lo; A.k = p.k; A.logL = p.logL; A.dif = 1; A.canonical = 0;
        ZK_TRY(launch_pass(s, st, A, false));
        A.src = nullptr;
        A.src2 = nullptr;
        A.src3 = nullptr;
    }
    {
        const PassPlan& p = passes[0];
        A.tw = dom->tw29_inv; A.tw2 = dom->tw29; A.bit_lo = 0; A.k = p.k; A.logL = 0; A.dif = 1; A.post = mid;
        A.canonical = (npass == 1) ? 1 : 0;
        unsigned E = 1u << p.k;
        size_t tiles = ((size_t)1 << logn) / E;
        if (g_ntt_g2) ZK_LAUNCH(s, st, "ntt_pass_contig_if", (k_ntt_pass29_if<2, 512>), dim3((unsigned)tiles, ny), dim3(512), (size_t)E * 36, A);
        else ZK_LAUNCH(s, st, "ntt_pass_contig_if", (k_ntt_pass29_if<3, 256>), dim3((unsigned)tiles, ny), dim3(NTT_THREADS), (size_t)E * 36, A);
        A.post = nullptr;
        A.tw2 = nullptr;
        A.src = nullptr;
        A.src2 = nullptr;
        A.data3 = nullptr;  // c is done
        A.src3 = nullptr;
    }
    for (size_t idx = 1; idx < npass; idx++) {  // strided passes of the forward transform, low bits first
        const PassPlan& p = passes[idx];
        A.tw = dom->tw29; A.bit_lo = p.bit_lo; A.k = p.k; A.logL = p.logL; A.dif = 0; A.canonical = (idx + 1 == npass) ? 1 : 0;
        ZK_TRY(launch_pass(s, st, A, false));
    }
    return ZK_OK;
}

static std::mutex g_lds_mu;
static uint64_t g_lds_attr_set = 0;  // bit e: done for device entry e (the attribute belongs to the device's copy of the function)
static int ensure_lds_attr() {
    std::lock_guard<std::mutex> lk(g_lds_mu);
    const uint64_t bit = (uint64_t)1 << current_entry();
    if (!(g_lds_attr_set & bit)) {
        ZK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ntt_pass), hipFuncAttributeMaxDynamicSharedMemorySize, (1 << TILE_LOG) * 32));
        ZK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ntt_pass29<3, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (1 << TILE_LOG) * 36));
        ZK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ntt_pass29<2, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, (1 << TILE_LOG) * 36));
        ZK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ntt_pass29<3, 256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (1 << TILE_LOG) * 36));
        ZK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ntt_pass29<2, 512, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (1 << TILE_LOG) * 36));
        ZK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ntt_pass29_if<3, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (1 << TILE_LOG) * 36));
        ZK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ntt_pass29_if<2, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, (1 << TILE_LOG) * 36));
        g_lds_attr_set |= bit;
    }
    return ZK_OK;
}

// (*Domain).FFT / FFTInverse on device memory
int ntt_dev(Slot* s, hipStream_t st, Fr* d_a, unsigned logn, int inverse, int decimation, int coset) {
    ZK_TRY(ensure_lds_attr());
    unsigned need = inverse ? DOM_TW_INV : DOM_TW;
    if (coset) {
        if (!inverse) need |= (decimation == ZK_DIT) ? DOM_COSET_REV : DOM_COSET;
        else need |= (decimation == ZK_DIT) ? DOM_COSET_INV_N : DOM_COSET_INV_N_REV;
    }
    Domain* d;
    ZK_TRY(get_domain(s, st, logn, need, &d));
    int dif = (decimation == ZK_DIF);
    if (!inverse) {
        // FFT: coset scaling first -- DIF by CosetTable[i], DIT (bit-reversed memory order) by CosetTableReversed[i]
        const Fr* pre = coset ? (dif ? d->coset_tab : d->coset_rev) : nullptr;
        return run_passes(s, st, d_a, d, 0, dif, pre, nullptr, nullptr);
    }
    // FFTInverse: TwiddlesInv, then * CardinalityInv (and CosetTableInv[i] for DIT / CosetTableInvReversed[i] for DIF)
    const Fr* post = coset ? (dif ? d->coset_inv_n_rev : d->coset_inv_n) : nullptr;
    Fr cinv = to_dev(d->card_inv);
    return run_passes(s, st, d_a, d, 1, dif, nullptr, post, &cinv);
}

int bit_reverse_dev(Slot* s, hipStream_t st, Fr* d_a, unsigned logn) {
    size_t n = (size_t)1 << logn;
    ZK_LAUNCH(s, st, "bit_reverse", k_bit_reverse, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d_a, logn);
    return ZK_OK;
}

// gnark v0.8.0 computeH on device buffers a (in/out, N), b, c (scratch, N); result left in a, bit-reversed order.
//   3 x FFTInverse(DIF) ; 3 x FFT(DIT, coset) ; a = (a*b - c) / (g^N - 1) ; FFTInverse(a, DIF, coset)
// The 1/N of the first inverse and the coset pre-scale g^bitrev(i) of the following forward transform are one
// table (coset_rev_n) applied while the inverse transform stores its last stage.
int compute_h_inplace(Slot* s, hipStream_t st, Fr* a, Fr* b, Fr* c, unsigned logN, const Fr* const* src, const hipStream_t* side) {
    // src (optional): the three inputs live in src[0..2] (full 2^logN vectors) and are left untouched; a, b, c are then pure outputs / scratch
    // side (optional): two more streams -- the transforms of b and c run on them, concurrently with a's on `st` (whatever produced the inputs must
    // be ordered before `st`): every pass is load -> butterflies -> store in lock-step over the whole machine at sizes that fit one round of
    // workgroups, so three transforms in flight put one array's loads / stores under another's butterflies.
    ZK_TRY(ensure_lds_attr());
    Domain* d;
    ZK_TRY(get_domain(s, st, logN, DOM_TW | DOM_TW_INV | DOM_COSET_REV_N | DOM_COSET_INV_N_REV, &d));
    size_t N = (size_t)1 << logN;
    Fr* vs[3] = {a, b, c};
    // By linearity (exact field arithmetic, so bit for bit for ANY input): FFTInverse(coset)((a'b' - c') den) = den (FFTInverse(coset)(a'b') - FFTInverse(c)),
    // where c' = FFT(coset)(FFTInverse(c)) -- the coset transform of c and its way back cancel.  c therefore only needs its first FFTInverse(DIF)
    // (coefficients, bit-reversed like the result): six transforms instead of gnark's seven.  ZKMI_H_SKIP_C=0 restores the literal sequence.
    static const bool skip_c = (ZK_EXP("ZKMI_H_SKIP_C", 1) != 0);
    if (skip_c && !side && logN > 0) {
        const Fr cinv = to_dev(d->card_inv);
        // a, b (and c for the inverse half) go through their passes in the SAME launches (gridDim.y = vector): at 2^20 a pass is ONE round of workgroups moving
        // in lock-step (load, butterflies, store); with two or three rounds per launch one round's loads and stores run under another's butterflies, and seven
        // launches disappear.  ZKMI_H_BATCH=0 (A/B switch): one vector per launch.
        static const bool h_batch = ZK_EXP("ZKMI_H_BATCH", 1) == 1;
        if (h_batch) {
            ZK_TRY(run_inverse_forward(s, st, a, d, d->coset_rev_n, src ? src[0] : nullptr, b, src ? src[1] : nullptr, c, src ? src[2] : nullptr, &cinv));
        } else {
            for (int i = 0; i < 2; i++) ZK_TRY(run_inverse_forward(s, st, vs[i], d, d->coset_rev_n, src ? src[i] : nullptr));
            ZK_TRY(run_passes(s, st, c, d, 1, 1, nullptr, nullptr, &cinv, src ? src[2] : nullptr));
        }
        HFr gN = d->coset;
        for (unsigned i = 0; i < logN; i++) gN = gN.sqr();
        const Fr den = to_dev((gN - HFr::one()).inv());
        // the product a*b rides on the loads of the closing transform's first stage (`pre` = b) and (x - c) * den on the stores of its last one (`sub` = c)
        // instead of two element-wise kernels (96 B per element each).  ZKMI_H_FUSE_PW=0 (A/B switch): the two kernels.
        static const bool fuse_pw = ZK_EXP("ZKMI_H_FUSE_PW", 1) == 1;
        if (fuse_pw && !g_ntt_saturated) return run_passes(s, st, a, d, 1, 1, b, d->coset_inv_n_rev, &den, nullptr, c);
        ZK_LAUNCH(s, st, "fr_mul", k_fr_mul, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, a, (const Fr*)a, (const Fr*)b, N);
        ZK_TRY(run_passes(s, st, a, d, 1, 1, nullptr, d->coset_inv_n_rev, nullptr));
        ZK_LAUNCH(s, st, "h_final", k_h_final, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, a, (const Fr*)c, den, N);
        return ZK_OK;
    }
    if (side) {
        hipEvent_t fork, join[2];
        ZK_HIP(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        ZK_HIP(hipEventRecord(fork, st));
        for (int i = 0; i < 2; i++) {
            ZK_HIP(hipStreamWaitEvent(side[i], fork, 0));
            ZK_TRY(run_inverse_forward(s, side[i], vs[i + 1], d, d->coset_rev_n, src ? src[i + 1] : nullptr));
            ZK_HIP(hipEventCreateWithFlags(&join[i], hipEventDisableTiming));
            ZK_HIP(hipEventRecord(join[i], side[i]));
        }
        ZK_TRY(run_inverse_forward(s, st, vs[0], d, d->coset_rev_n, src ? src[0] : nullptr));
        for (int i = 0; i < 2; i++) {
            ZK_HIP(hipStreamWaitEvent(st, join[i], 0));
            (void)hipEventDestroy(join[i]);
        }
        (void)hipEventDestroy(fork);
    } else {
        for (int i = 0; i < 3; i++) ZK_TRY(run_inverse_forward(s, st, vs[i], d, d->coset_rev_n, src ? src[i] : nullptr));
    }
    // den = 1 / (g^N - 1)
    HFr gN = d->coset;
    for (unsigned i = 0; i < logN; i++) gN = gN.sqr();
    HFr den = (gN - HFr::one()).inv();
    ZK_LAUNCH(s, st, "h_pointwise", k_h_pointwise, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, a, (const Fr*)b, (const Fr*)c, to_dev(den), N);
    ZK_TRY(run_passes(s, st, a, d, 1, 1, nullptr, d->coset_inv_n_rev, nullptr));
    return ZK_OK;
}

int fr_mul_dev(Slot* s, hipStream_t st, Fr* out, const Fr* a, const Fr* b, size_t n) {
    ZK_LAUNCH(s, st, "fr_mul", k_fr_mul, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, out, a, b, n);
    return ZK_OK;
}


// ------------------------------------------------------------------------------------ computeH sharded over G = 2^g GPUs
// Rank rho owns the block [rho*M, (rho+1)*M) of every length-D array (M = D/G).  A radix-2 transform over D splits into
//   * g CROSS stages on the top g index bits (the first g stages of a DIF transform, the last g of a DIT transform), and
//   * a plain size-M transform of the block (omega_M = omega_D^G), which is the single-GPU code above.
// For the cross stages the ranks transpose (all-to-all, done by the host over RCCL): rank rho then holds T[s][j], s < G,
// j < C = M/G, = element s*M + rho*C + j, and one lane runs the g butterflies of column j in registers (G <= 8: the
// radix-8 of SURVEY.md 8e).  Twiddles come from the size-D table: DIF stage t pairs s, s + (G >> (t+1)) with
// w^(((s mod h)*M + rho*C + j) << t); DIT cross stage u pairs s, s + 2^u with w^(((s mod 2^u)*M + rho*C + j) * (G >> (u+1))).
template <int LOGG, bool DIF>
__global__ __launch_bounds__(256) void k_ntt_cross(Fr* __restrict__ T, const Fr* __restrict__ tw, unsigned logM, unsigned rank) {
    prio_mid();
    constexpr unsigned G = 1u << LOGG;
    const size_t C = ((size_t)1 << logM) >> LOGG;
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= C) return;
    const size_t low = (size_t)rank * C + j;
    Fr x[G];
#pragma unroll
    for (unsigned s = 0; s < G; s++) x[s] = gload_fr(T + s * C + j);
#pragma unroll
    for (int st = 0; st < LOGG; st++) {
        const unsigned h = DIF ? (G >> (st + 1)) : (1u << st);  // pair distance in s
        const unsigned mul_shift = DIF ? st : (LOGG - 1 - st);  // exponent multiplier 2^t (DIF) / G/(2h) (DIT)
#pragma unroll
        for (unsigned s = 0; s < G; s++) {
            if (s & h) continue;
            const size_t e = (((size_t)(s & (h - 1)) << logM) + low) << mul_shift;
            Fr w = gload_fr(tw + e);
            if (DIF) {
                Fr u = x[s] + x[s + h];
                x[s + h] = (x[s] - x[s + h]) * w;
                x[s] = u;
            } else {
                Fr v = x[s + h] * w;
                x[s + h] = x[s] - v;
                x[s] = x[s] + v;
            }
        }
    }
#pragma unroll
    for (unsigned s = 0; s < G; s++) T[s * C + j] = x[s];
}

template <bool DIF>
static int launch_cross(Slot* s, hipStream_t st, Fr* T, const Fr* tw, unsigned logM, unsigned logg, unsigned rank) {
    const size_t C = ((size_t)1 << logM) >> logg;
    dim3 grid((unsigned)((C + 255) / 256)), block(256);
    switch (logg) {
        case 0: return ZK_OK;
        case 1: ZK_LAUNCH(s, st, "ntt_cross", (k_ntt_cross<1, DIF>), grid, block, 0, T, tw, logM, rank); return ZK_OK;
        case 2: ZK_LAUNCH(s, st, "ntt_cross", (k_ntt_cross<2, DIF>), grid, block, 0, T, tw, logM, rank); return ZK_OK;
        case 3: ZK_LAUNCH(s, st, "ntt_cross", (k_ntt_cross<3, DIF>), grid, block, 0, T, tw, logM, rank); return ZK_OK;
        default: return set_err(ZK_ERR_ARG, "computeH shards over at most 8 ranks (log_g = %u)", logg);
    }
}

struct ShardTables {
    Fr *coset_rev_n = nullptr, *coset_inv_n_rev = nullptr;  // this rank's M-entry slices of the size-D tables
    Fr *coset_nat = nullptr, *coset_rev = nullptr, *coset_inv_n_nat = nullptr;  // standalone sharded transforms (lazily): g^i, g^bitrev(i), g^-i / D
};
static std::mutex g_shard_mu;
static std::map<uint64_t, ShardTables> g_shard_tables;

static int get_shard_tables(Slot* s, hipStream_t st, Domain* dD, unsigned logD, unsigned logg, unsigned rank, ShardTables* out) {
    std::lock_guard<std::mutex> lk(g_shard_mu);
    const uint64_t key = ((uint64_t)current_entry() << 48) | ((uint64_t)logD << 32) | ((uint64_t)logg << 16) | rank;
    ShardTables& t = g_shard_tables[key];
    if (!t.coset_rev_n) {
        const size_t M = ((size_t)1 << logD) >> logg;
        ZK_TRY(make_pow_table(s, st, &t.coset_rev_n, M, logD, dD->coset, dD->card_inv, 1, (size_t)rank * M));
        ZK_TRY(make_pow_table(s, st, &t.coset_inv_n_rev, M, logD, dD->coset_inv, dD->card_inv, 1, (size_t)rank * M));
        ZK_HIP(hipStreamSynchronize(st));
    }
    *out = t;
    return ZK_OK;
}

// One compute phase of the sharded computeH (the host exchanges between phases; see include/zkmi.h).
int compute_h_shard_phase(Slot* s, hipStream_t st, int phase, Fr* a, Fr* b, Fr* c, unsigned logD, unsigned logg, unsigned rank) {
    if (logg > 3 || logD < 2 * logg || logD > 28) return set_err(ZK_ERR_ARG, "bad shard geometry (log_D = %u, log_g = %u)", logD, logg);
    if (rank >= (1u << logg)) return set_err(ZK_ERR_ARG, "rank %u out of range", rank);
    ZK_TRY(ensure_lds_attr());
    const unsigned logM = logD - logg;
    const size_t M = (size_t)1 << logM;
    Domain *dD, *dM;
    ZK_TRY(get_domain(s, st, logD, logg ? (DOM_TW | DOM_TW_INV) : 0u, &dD));
    ZK_TRY(get_domain(s, st, logM, DOM_TW | DOM_TW_INV, &dM));
    ShardTables tb;
    ZK_TRY(get_shard_tables(s, st, dD, logD, logg, rank, &tb));
    switch (phase) {
        case 0:  // transposed a, b, c (any subset: null arrays are skipped): cross stages of FFTInverse(DIF)
            for (Fr* v : {a, b, c})
                if (v) ZK_TRY(launch_cross<true>(s, st, v, dD->tw_inv, logM, logg, rank));
            return ZK_OK;
        case 1:  // blocks (any subset): rest of FFTInverse(DIF), * 1/D * g^bitrev(i), block part of FFT(DIT, coset)
            for (Fr* v : {a, b, c})
                if (v) ZK_TRY(run_inverse_forward(s, st, v, dM, tb.coset_rev_n));
            return ZK_OK;
        case 4:  // first third of phase 2, per array: cross stages of FFT(DIT) on the transposed arrays given
            for (Fr* v : {a, b, c})
                if (v) ZK_TRY(launch_cross<false>(s, st, v, dD->tw, logM, logg, rank));
            return ZK_OK;
        case 2:    // transposed: cross stages of FFT(DIT); pointwise; cross stages of the final FFTInverse(DIF, coset)
        case 5: {  // the rest of phase 2 after phase 4 ran on a, b, c: pointwise + cross stages of the final FFTInverse(DIF, coset)
            if (!a || !b || !c) return set_err(ZK_ERR_ARG, "phase %d needs a, b and c", phase);
            if (phase == 2)
                for (Fr* v : {a, b, c}) ZK_TRY(launch_cross<false>(s, st, v, dD->tw, logM, logg, rank));
            HFr gN = dD->coset;
            for (unsigned i = 0; i < logD; i++) gN = gN.sqr();
            HFr den = (gN - HFr::one()).inv();
            ZK_LAUNCH(s, st, "h_pointwise", k_h_pointwise, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, a, (const Fr*)b, (const Fr*)c, to_dev(den), M);
            return launch_cross<true>(s, st, a, dD->tw_inv, logM, logg, rank);
        }
        case 3:  // block of a: rest of FFTInverse(DIF, coset) -> this rank's block of h (gnark's bit-reversed order)
            return run_passes(s, st, a, dM, 1, 1, nullptr, tb.coset_inv_n_rev, nullptr);
        // ---- the six-transform schedule (c stays in coefficient form: compute_h_inplace's shortcut, sharded; one transpose and two transforms of c fewer)
        case 6: {  // blocks (any subset; the schedule uses it for c): rest of FFTInverse(DIF) with 1/D -> block of the coefficients, bit-reversed order
            const Fr cinv = to_dev(dD->card_inv);
            for (Fr* v : {a, b, c})
                if (v) ZK_TRY(run_passes(s, st, v, dM, 1, 1, nullptr, nullptr, &cinv));
            return ZK_OK;
        }
        case 7:  // transposed a, b (after phase 4 on each): a = a * b, then the cross stages of the final FFTInverse(DIF, coset) on a
            if (!a || !b) return set_err(ZK_ERR_ARG, "phase 7 needs a and b");
            ZK_LAUNCH(s, st, "fr_mul", k_fr_mul, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, a, (const Fr*)a, (const Fr*)b, M);
            return launch_cross<true>(s, st, a, dD->tw_inv, logM, logg, rank);
        case 8: {  // block of a and block of c's coefficients (phase 6): rest of FFTInverse(DIF, coset), then h = (a - c) / (g^D - 1)
            if (!a || !c) return set_err(ZK_ERR_ARG, "phase 8 needs a and c");
            ZK_TRY(run_passes(s, st, a, dM, 1, 1, nullptr, tb.coset_inv_n_rev, nullptr));
            HFr gN = dD->coset;
            for (unsigned i = 0; i < logD; i++) gN = gN.sqr();
            const HFr den = (gN - HFr::one()).inv();
            ZK_LAUNCH(s, st, "h_final", k_h_final, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, a, (const Fr*)c, to_dev(den), M);
            return ZK_OK;
        }
        default: return set_err(ZK_ERR_ARG, "phase must be 0..8");
    }
}

// One step of a standalone transform sharded by blocks over G = 2^g ranks (BASELINE configs[4] on several GPUs; the same decomposition as computeH's):
// (*Domain).FFT / FFTInverse over D = G * M points, rank rho holding block rho of the STORED order (natural for DIF input / DIT output, bit-reversed
// for DIF output / DIT input).  The host sequences the steps with its all-to-all transposes (X):
//     FFT(DIF):         [SCALE if coset]  X  CROSS  X  BLOCK             FFT(DIT):         BLOCK  X  CROSS  X
//     FFTInverse(DIF):                    X  CROSS  X  BLOCK             FFTInverse(DIT):  BLOCK  X  CROSS  X  [SCALE if coset]
//   step 0 CROSS: the g stages on the top index bits, on transposed data;  step 1 BLOCK: the size-M transform of the block with whatever scaling
//   can ride on it (1/D always; g^bitrev(i) before a forward DIT, g^-bitrev(i) after an inverse DIF);  step 2 SCALE: the coset factor that must act on
//   the natural-order block outside the block transform (g^i before a forward DIF's cross stages, g^-i after an inverse DIT's).
int ntt_shard_step(Slot* s, hipStream_t st, int step, Fr* a, unsigned logD, unsigned logg, unsigned rank, int inverse, int dif, int coset) {
    if (logg > 3 || logD < 2 * logg || logD > 28) return set_err(ZK_ERR_ARG, "bad shard geometry (log_D = %u, log_g = %u)", logD, logg);
    if (rank >= (1u << logg)) return set_err(ZK_ERR_ARG, "rank %u out of range", rank);
    ZK_TRY(ensure_lds_attr());
    const unsigned logM = logD - logg;
    const size_t M = (size_t)1 << logM;
    Domain *dD, *dM;
    ZK_TRY(get_domain(s, st, logD, logg ? (DOM_TW | DOM_TW_INV) : 0u, &dD));
    ZK_TRY(get_domain(s, st, logM, DOM_TW | DOM_TW_INV, &dM));
    ShardTables tb;
    ZK_TRY(get_shard_tables(s, st, dD, logD, logg, rank, &tb));
    if (coset) {
        std::lock_guard<std::mutex> lk(g_shard_mu);
        ShardTables& t = g_shard_tables[((uint64_t)current_entry() << 48) | ((uint64_t)logD << 32) | ((uint64_t)logg << 16) | rank];  // the key of get_shard_tables
        if (!t.coset_nat) {
            ZK_TRY(make_pow_table(s, st, &t.coset_nat, M, logD, dD->coset, HFr::one(), 0, (size_t)rank * M));
            ZK_TRY(make_pow_table(s, st, &t.coset_rev, M, logD, dD->coset, HFr::one(), 1, (size_t)rank * M));
            ZK_TRY(make_pow_table(s, st, &t.coset_inv_n_nat, M, logD, dD->coset_inv, HFr::one(), 0, (size_t)rank * M));
            ZK_HIP(hipStreamSynchronize(st));
        }
        tb = t;
    }
    const Fr cinv = to_dev(dD->card_inv);  // 1/D: the block transform is linear, so the whole scaling rides on it
    switch (step) {
        case 0:
            return dif ? launch_cross<true>(s, st, a, inverse ? dD->tw_inv : dD->tw, logM, logg, rank)
                       : launch_cross<false>(s, st, a, inverse ? dD->tw_inv : dD->tw, logM, logg, rank);
        case 1:
            if (!inverse) return run_passes(s, st, a, dM, 0, dif, (coset && !dif) ? tb.coset_rev : nullptr, nullptr, nullptr);
            return run_passes(s, st, a, dM, 1, dif, nullptr, (coset && dif) ? tb.coset_inv_n_rev : nullptr, &cinv);
        case 2: {
            if (!coset || (!inverse && !dif) || (inverse && dif)) return ZK_OK;  // nothing left outside the block transform
            const Fr* t = inverse ? tb.coset_inv_n_nat : tb.coset_nat;
            ZK_LAUNCH(s, st, "ntt_scale", k_scale_table, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, a, t, M);
            return ZK_OK;
        }
        default: return set_err(ZK_ERR_ARG, "step must be 0 (cross), 1 (block) or 2 (scale)");
    }
}

}  // namespace zkmi

using namespace zkmi;

extern "C" {

int zk_bn254_ntt_dev(void* d_a, uint32_t log_n, int inverse, int decimation, int coset, void* stream) {
    if (!d_a) return set_err(ZK_ERR_ARG, "null data pointer");
    if (log_n > 28) return set_err(ZK_ERR_ARG, "log_n = %u exceeds Fr two-adicity 28", log_n);
    if (decimation != ZK_DIT && decimation != ZK_DIF) return set_err(ZK_ERR_ARG, "decimation must be ZK_DIT or ZK_DIF");
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = stream ? (hipStream_t)stream : g.s->stream;
    ZK_TRY(ntt_dev(g.s, st, (Fr*)d_a, log_n, inverse, decimation, coset));
    // no workspace is used: safe to return without synchronising a caller-provided stream
    if (!stream || profiling_on()) ZK_TRY(slot_sync(g.s, st));
    return ZK_OK;
}

// device_mask: bit i = device entry i takes a block of the array (2, 4 or 8 entries: multidev.hip); 0 = the process default, which only spreads transforms
// of at least 2^18 points per entry
int zk_bn254_ntt_devices(zk_fr* a, uint32_t log_n, int inverse, int decimation, int coset, uint32_t device_mask) {
    if (!a) return set_err(ZK_ERR_ARG, "null data pointer");
    if (log_n > 28) return set_err(ZK_ERR_ARG, "log_n = %u exceeds Fr two-adicity 28", log_n);
    if (decimation != ZK_DIT && decimation != ZK_DIF) return set_err(ZK_ERR_ARG, "decimation must be ZK_DIT or ZK_DIF");
    std::vector<int> ents;
    ZK_TRY(md_entries_for(device_mask, (size_t)1 << log_n, (size_t)1 << 18, &ents));
    if (ents.size() > 1 && (device_mask || ents.size() == 2 || ents.size() == 4 || ents.size() == 8)) return md_ntt_host(a, log_n, inverse, decimation, coset, ents);
    CtxScope sc(ents[0]);
    if (sc.rc != ZK_OK) return sc.rc;
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    size_t bytes = ((size_t)1 << log_n) * 32;
    ZK_TRY(g.s->reserve(bytes));
    Fr* d = (Fr*)g.s->alloc(bytes);
    hipStream_t st = g.s->stream;
    ZK_HIP(hipMemcpyAsync(d, a, bytes, hipMemcpyHostToDevice, st));
    ZK_TRY(ntt_dev(g.s, st, d, log_n, inverse, decimation, coset));
    ZK_HIP(hipMemcpyAsync(a, d, bytes, hipMemcpyDeviceToHost, st));
    return slot_sync(g.s, st);
}
int zk_bn254_ntt(zk_fr* a, uint32_t log_n, int inverse, int decimation, int coset) { return zk_bn254_ntt_devices(a, log_n, inverse, decimation, coset, 0); }

int zk_bn254_bit_reverse_dev(void* d_a, uint32_t log_n, void* stream) {
    if (!d_a || log_n > 28) return set_err(ZK_ERR_ARG, "bad argument");
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = stream ? (hipStream_t)stream : g.s->stream;
    ZK_TRY(bit_reverse_dev(g.s, st, (Fr*)d_a, log_n));
    if (!stream || profiling_on()) ZK_TRY(slot_sync(g.s, st));
    return ZK_OK;
}

int zk_bn254_bit_reverse(zk_fr* a, uint32_t log_n) {
    if (!a || log_n > 28) return set_err(ZK_ERR_ARG, "bad argument");
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    size_t bytes = ((size_t)1 << log_n) * 32;
    ZK_TRY(g.s->reserve(bytes));
    Fr* d = (Fr*)g.s->alloc(bytes);
    hipStream_t st = g.s->stream;
    ZK_HIP(hipMemcpyAsync(d, a, bytes, hipMemcpyHostToDevice, st));
    ZK_TRY(bit_reverse_dev(g.s, st, d, log_n));
    ZK_HIP(hipMemcpyAsync(a, d, bytes, hipMemcpyDeviceToHost, st));
    return slot_sync(g.s, st);
}

int zk_bn254_groth16_compute_h_dev(const void* d_a, const void* d_b, const void* d_c, size_t n, uint32_t log_N, void* d_h_out, void* stream) {
    if (!d_a || !d_b || !d_c || !d_h_out) return set_err(ZK_ERR_ARG, "null pointer");
    if (log_N > 28) return set_err(ZK_ERR_ARG, "log_N exceeds 28");
    size_t N = (size_t)1 << log_N;
    if (n > N) return set_err(ZK_ERR_ARG, "n = %zu exceeds the domain size %zu", n, N);
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = stream ? (hipStream_t)stream : g.s->stream;
    ZK_TRY(g.s->reserve(2 * N * 32 + 512));
    Fr* a = (Fr*)d_h_out;
    Fr* b = (Fr*)g.s->alloc(N * 32);
    Fr* c = (Fr*)g.s->alloc(N * 32);
    const void* src[3] = {d_a, d_b, d_c};
    Fr* dst[3] = {a, b, c};
    for (int i = 0; i < 3; i++) {
        if (src[i] != dst[i]) ZK_HIP(hipMemcpyAsync(dst[i], src[i], n * 32, hipMemcpyDeviceToDevice, st));
        if (n < N) ZK_HIP(hipMemsetAsync(dst[i] + n, 0, (N - n) * 32, st));
    }
    ZK_TRY(compute_h_inplace(g.s, st, a, b, c, log_N));
    return slot_sync(g.s, st);  // scratch lives in the slot's arena
}

int zk_bn254_groth16_compute_h(const zk_fr* a, const zk_fr* b, const zk_fr* c, size_t n, uint32_t log_N, zk_fr* h_out) {
    if (!a || !b || !c || !h_out) return set_err(ZK_ERR_ARG, "null pointer");
    if (log_N > 28) return set_err(ZK_ERR_ARG, "log_N exceeds 28");
    size_t N = (size_t)1 << log_N;
    if (n > N) return set_err(ZK_ERR_ARG, "n = %zu exceeds the domain size %zu", n, N);
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = g.s->stream;
    ZK_TRY(g.s->reserve(3 * N * 32 + 1024));
    Fr* d[3];
    const zk_fr* src[3] = {a, b, c};
    for (int i = 0; i < 3; i++) {
        d[i] = (Fr*)g.s->alloc(N * 32);
        ZK_HIP(hipMemcpyAsync(d[i], src[i], n * 32, hipMemcpyHostToDevice, st));
        if (n < N) ZK_HIP(hipMemsetAsync(d[i] + n, 0, (N - n) * 32, st));
    }
    ZK_TRY(compute_h_inplace(g.s, st, d[0], d[1], d[2], log_N));
    ZK_HIP(hipMemcpyAsync(h_out, d[0], N * 32, hipMemcpyDeviceToHost, st));
    return slot_sync(g.s, st);
}

int zk_bn254_groth16_h_shard_dev(int phase, void* d_a, void* d_b, void* d_c, uint32_t log_D, uint32_t log_g, uint32_t rank, void* stream) {
    if ((!d_a && phase != 6) || ((phase == 2 || phase == 5) && (!d_b || !d_c))) return set_err(ZK_ERR_ARG, "null pointer");
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    // Give it the stream of the proof's msm5 session (zk_bn254_groth16_msm5_session_stream): a foreign stream may share a
    // hardware queue with the stream that prepares the wire scalars of the same proof (measured with a torch-created stream:
    // the two serialised, ~0.5 ms per proof).
    hipStream_t st = stream ? (hipStream_t)stream : g.s->stream;
    ZK_TRY(compute_h_shard_phase(g.s, st, phase, (Fr*)d_a, (Fr*)d_b, (Fr*)d_c, log_D, log_g, rank));
    if (!stream) ZK_TRY(slot_sync(g.s, st));  // in place, no workspace: a caller-provided stream stays asynchronous (profiled or not)
    return ZK_OK;
}

int zk_bn254_ntt_shard_dev(int step, void* d_a, uint32_t log_D, uint32_t log_g, uint32_t rank, int inverse, int decimation, int coset, void* stream) {
    if (!d_a) return set_err(ZK_ERR_ARG, "null data pointer");
    if (decimation != ZK_DIT && decimation != ZK_DIF) return set_err(ZK_ERR_ARG, "decimation must be ZK_DIT or ZK_DIF");
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = stream ? (hipStream_t)stream : g.s->stream;
    ZK_TRY(ntt_shard_step(g.s, st, step, (Fr*)d_a, log_D, log_g, rank, inverse ? 1 : 0, decimation == ZK_DIF, coset ? 1 : 0));
    if (!stream) ZK_TRY(slot_sync(g.s, st));
    return ZK_OK;
}

int zk_bn254_fr_mul_dev(void* d_out, const void* d_a, const void* d_b, size_t n, void* stream) {
    if (!d_out || !d_a || !d_b) return set_err(ZK_ERR_ARG, "null pointer");
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = stream ? (hipStream_t)stream : g.s->stream;
    ZK_TRY(fr_mul_dev(g.s, st, (Fr*)d_out, (const Fr*)d_a, (const Fr*)d_b, n));
    if (!stream || profiling_on()) ZK_TRY(slot_sync(g.s, st));
    return ZK_OK;
}

}  // extern "C"
