// Decision benchmark ("next" #4a): would a 5 x 52-bit-limb Montgomery product on the FP64 pipe beat the 9 x 29-bit integer one (ff29.hpp:
// 162 v_mad_u64_u32 + 44 others, 918 cycles per wave)?  The construction is Emmart / Zheng / Weems' (ARITH 2018): for limbs a, b < 2^52 held as doubles
//     hi = fma(a, b, 2^104)            -- the top 52 bits of the 104-bit product, in the mantissa of a number with exponent 104
//     lo = fma(a, b, 2^104 + 2^52 - hi) -- the low 52 bits (+ 2^52), exact
// (round toward zero), and the column sums are taken on the BIT PATTERNS with 64-bit integer additions, the accumulated biases subtracted once per column.
// Per limb product: 2 v_fma_f64 + 1 v_add_f64 + 2 v_lshl_add_u64.  A Montgomery product is 25 limb products for a*b, 5 for the quotient digits (low halves only)
// and 25 for q*p, then ten columns to resolve into five limbs and convert back to doubles.
//
// This kernel executes exactly that instruction mix on dependent data (so nothing can be hoisted or dropped) and is timed next to u29_mul in the same
// run, at the same occupancies.  It is an OPTIMISTIC bound for the FP64 route: rounding-mode switches, the final conditional subtraction and the
// conversions at the kernel boundary are left out, and the result is not checked for arithmetic correctness -- only if the bound were clearly below 918
// cycles would a checked implementation be worth writing (the bar: < 780 cycles per wave).
//
//   hipcc -O3 --offload-arch=gfx950 -I../noir_backend_using_gnark_amd/csrc ubench4.hip -o ubench4 ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "ff.hpp"
#include "curve.hpp"
#include "ff29.hpp"
using namespace zkmi;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct D5 {
    double l[5];
};
__device__ __forceinline__ long long bits(double x) { return __double_as_longlong(x); }

// one Montgomery product in the Emmart form; p, ninv: modulus limbs and -p^-1 mod 2^52 as doubles
__device__ __forceinline__ D5 mont52(const D5& a, const D5& b, const D5& p, double ninv) {
    const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
    long long col[11];
#pragma unroll
    for (int k = 0; k < 11; k++) col[k] = 0;
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const double hi = __builtin_fma(a.l[i], b.l[j], C1);
            const double lo = __builtin_fma(a.l[i], b.l[j], C2 - hi);
            col[i + j + 1] += bits(hi);
            col[i + j] += bits(lo);
        }
    const long long MASK = (1LL << 52) - 1;
    long long carry = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        // quotient digit: low 52 bits of (column value * ninv); the column's biases are compile-time constants folded into the subtraction
        const long long c = col[i] + carry - (long long)(i + 1) * 0x4330000000000000LL - (long long)i * 0x4670000000000000LL;
        const double cl = (double)(c & MASK);
        const double qh = __builtin_fma(cl, ninv, C1);
        const double ql = __builtin_fma(cl, ninv, C2 - qh);
        const double q = __longlong_as_double((bits(ql) & MASK) | 0x4330000000000000LL) - 0x1p52;  // low 52 bits back as a double
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const double hi = __builtin_fma(q, p.l[j], C1);
            const double lo = __builtin_fma(q, p.l[j], C2 - hi);
            col[i + j + 1] += bits(hi);
            col[i + j] += bits(lo);
        }
        carry = (col[i] + carry) >> 52;
    }
    D5 r;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const long long c = col[5 + i] + carry - 10LL * 0x4330000000000000LL;
        carry = c >> 52;
        r.l[i] = __longlong_as_double((c & MASK) | 0x4330000000000000LL) - 0x1p52;
    }
    return r;
}

__global__ void k_mont52(double* out, const double* in, int iters) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    D5 a, b, p;
#pragma unroll
    for (int i = 0; i < 5; i++) { a.l[i] = in[t * 5 + i]; b.l[i] = in[t * 5 + i] + 3.0; p.l[i] = in[i] + 7.0; }
    const double ninv = in[5] + 11.0;
    for (int it = 0; it < iters; it++) {
        a = mont52(a, b, p, ninv);
        b = mont52(b, a, p, ninv);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) s += a.l[i] + b.l[i];
    out[t] = s;
}
__global__ void k_mul29(Fp* out, const Fp* in, int iters) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    U29 a = u29_load(in[t]), b = u29_load(in[t + 1]);
    for (int it = 0; it < iters; it++) {
        a = u29_mul(a, b);
        b = u29_mul(b, a);
    }
    out[t] = u29_store(u29_add(a, b));
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
    double *din, *dout;
    Fp *fin, *fout;
    CHECK(hipMalloc(&din, nmax * 5 * 8));
    CHECK(hipMalloc(&dout, nmax * 8));
    CHECK(hipMalloc(&fin, nmax * sizeof(Fp)));
    CHECK(hipMalloc(&fout, nmax * sizeof(Fp)));
    std::vector<double> h(nmax * 5);
    for (size_t i = 0; i < h.size(); i++) h[i] = (double)((i * 2654435761ull) & ((1ull << 51) - 1));
    CHECK(hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    std::vector<uint32_t> hf(nmax * 8);
    for (size_t i = 0; i < hf.size(); i++) hf[i] = (uint32_t)(i * 2654435761u) & ((i % 8 == 7) ? 0x0fffffffu : 0xffffffffu);
    CHECK(hipMemcpy(fin, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
    printf("{\"device\": \"%s\", \"cus\": %d, \"rows\": [\n", prop.name, cus);
    bool first = true;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = cus * wps, it = 400;
        const double ms52 = timeit(blocks, 256, k_mont52, dout, (const double*)din, it);
        const double ms29 = timeit(blocks, 256, k_mul29, fout, (const Fp*)fin, it);
        const double c52 = ms52 * 1e-3 * 2.4e9 / (it * 2.0 * wps), c29 = ms29 * 1e-3 * 2.4e9 / (it * 2.0 * wps);
        printf("%s {\"waves_per_simd\": %d, \"fp64_5x52_cycles_per_wave_product\": %.1f, \"int_9x29_cycles_per_wave_product\": %.1f, \"fp64_over_int\": %.3f}", first ? "" : ",\n", wps, c52,
               c29, c52 / c29);
        first = false;
    }
    printf("\n]}\n");
    return 0;
}
