// Decision benchmark ("next" #5): does the CONSTANT-operand half of a Montgomery product belong on the matrix cores?
//
// In  r = (t + m q) / R,  t = a b,  m = (t mod R) q' mod R  two of the three big products have a constant operand (q' and q).  For the 64 elements of a wave,
// m q is a dense contraction: with m and q in bytes, column k of the product is sum_j m_j q_(k-j) -- a [columns x bytes] Toeplitz(q) matrix times a
// [bytes x elements] matrix of the lanes' m, i.e. int8 MFMA work (v_mfma_i32_32x32x32_i8; published for CUDA tensor cores: DistMSM, ASPLOS '24) that issues beside
// the VALU instead of on it.  north_star says "MFMA is not used (no dense contraction)"; this tests that premise against today's schedule
// (ff29.hpp u29_mul: 162 v_mad_u64_u32 + 44 others, all VALU).
//
// Layout that makes it possible at all.  D = Toeplitz(q) [M = columns] x m_bytes [K = 32 bytes][N = elements]: the B operand of a 32x32x32 tile wants, in lane l,
// bytes (l / 32) * 16 .. + 15 of element l % 32, and the result leaves lane l with 16 of the 32 column sums of element l % 32 -- so a wave covers its 64 elements
// with two N-tiles, and lanes l and l + 32 trade halves with v_permlane32_swap: 4 swaps per 32 bytes going in, 16 per 32-column tile coming out.
// Only the HIGH half of m q is needed (the low half cancels t mod R by construction and contributes one carry bit), plus three guard columns to fix that
// carry: 2 M-tiles x 2 N-tiles = 4 MFMAs per wave and product (K = 32: R = 2^256 here; the 9 x 29-bit form's R = 2^261 would need a second K step).
//
// What is measured: an OPTIMISTIC BOUND, like tools/ubench4.hip -- the instruction mix of the MFMA route on dependent data (nothing can be hoisted or dropped),
// not a checked implementation:
//     81 v_mad_u64_u32           t = a b (9 x 9 limbs of 29 bits: unchanged, both operands vary)
//     45 v_mad_u64_u32 + masks   m = (t mod R) q' mod R   (lower triangle; constant operand, but its 32 byte columns would need a serial carry chain to become
//                                the bytes the next step consumes: not cheaper on the matrix cores)
//     ~20 shifts / ors           9 x 29-bit limbs of m -> 8 words of 4 bytes
//     4 v_permlane32_swap        B operands of the two N-tiles
//     4 v_mfma_i32_32x32x32_i8   the 64 column sums of the high half for 64 elements
//     32 v_permlane32_swap       every lane gets its own element's 64 sums
//     64 -> 9 limbs              the sums sit at BYTE spacing and are 21 bits wide: folding them into 29-bit limbs is one shifted multiply-add per column
//                                (v_mad_u64_u32 with a power of two: 64 of them) -- the step that decides the outcome: it costs what the 81 products it replaces cost
// Only if this bound were clearly below u29_mul's cycles per wave (the bar: <= 0.8 x) would a checked implementation be worth writing.
//
//   hipcc -O3 --offload-arch=gfx950 -I../noir_backend_using_gnark_amd/csrc ubench5.hip -o ubench5 ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "ff.hpp"
#include "curve.hpp"
#include "ff29.hpp"
using namespace zkmi;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct L9 {
    uint32_t l[9];
};
__device__ __forceinline__ void swap32(uint32_t& a, uint32_t& b) {  // a[32..63] <-> b[0..31]
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
#else
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
#endif
}

// the MFMA route's instruction mix for one product (see the header); tq: the Toeplitz rows of q as this lane's A operands (4 tiles), qp / q: limbs of q' and q
__device__ __forceinline__ L9 mul_mfma_route(const L9& a, const L9& b, const L9& qp, const v4i tq[4]) {
    const uint32_t M29 = (1u << 29) - 1;
    // t = a b: 17 columns, 81 products
    uint64_t col[18];
#pragma unroll
    for (int k = 0; k < 18; k++) col[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) col[i + j] += (uint64_t)a.l[i] * b.l[j];
    // carries of the low half, then m = t_lo q' mod R (lower triangle: 45 products)
    uint32_t tl[9];
    uint64_t c = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        c += col[k];
        tl[k] = (uint32_t)c & M29;
        c >>= 29;
    }
    uint64_t mc[9];
#pragma unroll
    for (int k = 0; k < 9; k++) mc[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; i + j < 9; j++) mc[i + j] += (uint64_t)tl[i] * qp.l[j];
    uint32_t m[9];
    uint64_t cm = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        cm += mc[k];
        m[k] = (uint32_t)cm & M29;
        cm >>= 29;
    }
    // 9 x 29 bits -> 8 words of 4 bytes (the top 5 bits are dropped: R = 2^256 in this bound)
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int bit = 32 * k, i0 = bit / 29, sh = bit % 29;
        uint32_t v = m[i0] >> sh;
        v |= m[i0 + 1] << (29 - sh);
        if (29 - sh + 29 < 32 && i0 + 2 < 9) v |= m[i0 + 2] << (58 - sh);
        w[k] = v;
    }
    // B operands: lanes l and l + 32 trade halves so that tile 0 holds elements 0..31 and tile 1 elements 32..63
    uint32_t b0[4] = {w[0], w[1], w[2], w[3]}, b1[4] = {w[4], w[5], w[6], w[7]};
#pragma unroll
    for (int k = 0; k < 4; k++) swap32(b0[k], b1[k]);
    const v4i B0 = {(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3]}, B1 = {(int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
    v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v16i d00 = __builtin_amdgcn_mfma_i32_32x32x32_i8(tq[0], B0, z, 0, 0, 0);  // columns 0..31, elements 0..31
    v16i d01 = __builtin_amdgcn_mfma_i32_32x32x32_i8(tq[1], B1, z, 0, 0, 0);  // columns 0..31, elements 32..63
    v16i d10 = __builtin_amdgcn_mfma_i32_32x32x32_i8(tq[2], B0, z, 0, 0, 0);  // columns 32..63
    v16i d11 = __builtin_amdgcn_mfma_i32_32x32x32_i8(tq[3], B1, z, 0, 0, 0);
    // every lane collects its own element's 64 sums: 16 swaps per column tile
    uint32_t s[64];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        uint32_t x = (uint32_t)d00[k], y = (uint32_t)d01[k];
        swap32(x, y);
        s[2 * k] = x;
        s[2 * k + 1] = y;
        uint32_t x2 = (uint32_t)d10[k], y2 = (uint32_t)d11[k];
        swap32(x2, y2);
        s[32 + 2 * k] = x2;
        s[32 + 2 * k + 1] = y2;
    }
    // fold the byte-spaced sums (21 bits each) into the high columns of t: one shifted multiply-add per column
    uint64_t hi[9];
#pragma unroll
    for (int k = 0; k < 9; k++) hi[k] = col[9 + k];
    hi[0] += c + cm;
#pragma unroll
    for (int k = 0; k < 64; k++) {
        const int bit = 8 * k, i0 = bit / 29, sh = bit % 29;
        hi[i0 < 9 ? i0 : 8] += (uint64_t)s[k] * (1u << sh);
    }
    L9 r;
    uint64_t cr = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        cr += hi[k];
        r.l[k] = (uint32_t)cr & M29;
        cr >>= 29;
    }
    r.l[8] += (uint32_t)cr << 29;
    return r;
}

__global__ __launch_bounds__(256) void k_mfma_route(uint32_t* out, const uint32_t* in, int iters) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    L9 a, b, qp;
    v4i tq[4];
#pragma unroll
    for (int i = 0; i < 9; i++) { a.l[i] = in[t * 9 + i] & 0x1fffffffu; b.l[i] = (in[t * 9 + i] * 2654435761u) & 0x1fffffffu; qp.l[i] = in[i] & 0x1fffffffu; }
#pragma unroll
    for (int k = 0; k < 4; k++) tq[k] = v4i{(int)in[16 + 4 * k], (int)in[17 + 4 * k], (int)in[18 + 4 * k], (int)in[19 + 4 * k]};
    for (int it = 0; it < iters; it++) {
        a = mul_mfma_route(a, b, qp, tq);
        b = mul_mfma_route(b, a, qp, tq);
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) s += a.l[i] ^ b.l[i];
    out[t] = s;
}
__global__ __launch_bounds__(256) void k_mul29(Fp* out, const Fp* in, int iters) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    U29 a = u29_load(in[t]), b = u29_load(in[t + 1]);
    for (int it = 0; it < iters; it++) {
        a = u29_mul(a, b);
        b = u29_mul(b, a);
    }
    out[t] = u29_store(u29_add(a, b));
}
// the pieces alone: how many cycles a wave pays for the swaps and for the MFMAs when nothing else is in flight
__global__ __launch_bounds__(256) void k_swaps(uint32_t* out, const uint32_t* in, int iters) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = in[t * 8 + i];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 9; r++)  // 36 swaps
#pragma unroll
            for (int i = 0; i < 4; i++) { swap32(x[i], x[4 + i]); x[i] += 1; }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i];
    out[t] = s;
}
__global__ __launch_bounds__(256) void k_mfmas(uint32_t* out, const uint32_t* in, int iters) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    v4i A = {(int)in[t], (int)in[t + 1], (int)in[t + 2], (int)in[t + 3]}, B = {(int)in[t + 4], (int)in[t + 5], (int)in[t + 6], (int)in[t + 7]};
    v16i acc[4];
#pragma unroll
    for (int k = 0; k < 4; k++) acc[k] = v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc[k], 0, 0, 0);  // four independent accumulators
    }
    int s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int i = 0; i < 16; i++) s += acc[k][i];
    out[t] = (uint32_t)s;
}

template <class K, class... A>
static double timeit(int blocks, int threads, K kern, A... args) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, args...);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, args...);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t nmax = (size_t)cus * 8 * 256 + 64;
    uint32_t *uin, *uout;
    Fp *fin, *fout;
    CHECK(hipMalloc(&uin, nmax * 9 * 4 + 4096));
    CHECK(hipMalloc(&uout, nmax * 4));
    CHECK(hipMalloc(&fin, nmax * sizeof(Fp)));
    CHECK(hipMalloc(&fout, nmax * sizeof(Fp)));
    std::vector<uint32_t> hu(nmax * 9 + 1024);
    for (size_t i = 0; i < hu.size(); i++) hu[i] = (uint32_t)(i * 2654435761u + 12345u);
    CHECK(hipMemcpy(uin, hu.data(), hu.size() * 4, hipMemcpyHostToDevice));
    std::vector<uint32_t> hf(nmax * 8);
    for (size_t i = 0; i < hf.size(); i++) hf[i] = (uint32_t)(i * 2654435761u) & ((i % 8 == 7) ? 0x0fffffffu : 0xffffffffu);
    CHECK(hipMemcpy(fin, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
    const double clk = 2.4e9;
    printf("{\"device\": \"%s\", \"cus\": %d, \"rows\": [\n", prop.name, cus);
    bool first = true;
    for (int wps : {1, 2, 4}) {
        const int blocks = cus * wps, it = 400;
        const double ms_m = timeit(blocks, 256, k_mfma_route, uout, (const uint32_t*)uin, it);
        const double ms_29 = timeit(blocks, 256, k_mul29, fout, (const Fp*)fin, it);
        const double ms_sw = timeit(blocks, 256, k_swaps, uout, (const uint32_t*)uin, it);
        const double ms_mf = timeit(blocks, 256, k_mfmas, uout, (const uint32_t*)uin, it);
        const double c_m = ms_m * 1e-3 * clk / (it * 2.0 * wps), c_29 = ms_29 * 1e-3 * clk / (it * 2.0 * wps);
        const double c_sw = ms_sw * 1e-3 * clk / (it * 36.0 * wps), c_mf = ms_mf * 1e-3 * clk / (it * 4.0 * wps);
        printf("%s {\"waves_per_simd\": %d, \"mfma_route_bound_cycles_per_wave_product\": %.1f, \"int_9x29_cycles_per_wave_product\": %.1f, \"mfma_route_over_int\": %.3f, "
               "\"cycles_per_permlane32_swap\": %.2f, \"cycles_per_mfma_i32_32x32x32_i8\": %.1f}",
               first ? "" : ",\n", wps, c_m, c_29, c_m / c_29, c_sw, c_mf);
        first = false;
    }
    printf("\n],\n \"bar\": \"mfma_route_over_int <= 0.80 at 4 waves per SIMD, else the route is closed (DESIGN.md 8)\"}\n");
    return 0;
}
