// Instruction-throughput micro-benchmarks for gfx950 that decide how the 254-bit modular multiply is built.
// build: hipcc -O3 --offload-arch=gfx950 -I../noir_backend_using_gnark_amd/csrc ubench.hip -o ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "ff.hpp"
#include "curve.hpp"
#include "ff29.hpp"
using namespace zkmi;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int ILP>
__global__ void k_mad64(uint64_t* out, uint32_t a, uint32_t b, int iters) {
    uint64_t acc[ILP];
    for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i;
    uint32_t x = a + threadIdx.x, y = b;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "vcc");
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_mullo(uint64_t* out, uint32_t a, uint32_t b, int iters) {
    uint32_t acc[ILP];
    for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i + a;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(b));
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_add32(uint64_t* out, uint32_t a, uint32_t b, int iters) {
    uint32_t acc[ILP];
    for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i + a;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %1, vcc" : "+v"(acc[i]) : "v"(b) : "vcc");
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_fma64(uint64_t* out, double a, double b, int iters) {
    double acc[ILP];
    for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
    }
    double s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}
template <int ILP>
__global__ void k_lshladd64(uint64_t* out, uint64_t a, int iters) {
    uint64_t acc[ILP];
    for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(a));
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_pkmov(uint64_t* out, uint64_t a, int iters) {
    uint64_t acc[ILP];
    for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) asm volatile("v_pk_mov_b32 %0, %0, %1 op_sel:[1,0]" : "+v"(acc[i]) : "v"(a));
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_modmul(Fp* out, const Fp* in, int iters) {
    size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    Fp a = in[i], b = in[i + 1];
    for (int it = 0; it < iters; it++) { a = a * b; b = b * a; }
    out[i] = a + b;
}
__global__ void k_modadd(Fp* out, const Fp* in, int iters) {
    size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    Fp a = in[i], b = in[i + 1];
    for (int it = 0; it < iters; it++) { a = a + b; b = b - a; }
    out[i] = a + b;
}
__global__ __launch_bounds__(256) void k_madd(G1XYZZ* out, const G1Affine* in, int iters) {
    size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    G1XYZZ acc = G1XYZZ::from_affine(in[i]);
    G1Affine p = in[i + 1];
    for (int it = 0; it < iters; it++) { acc.madd(p.x, p.y); p.x = p.x + acc.x; }
    out[i] = acc;
}

__global__ __launch_bounds__(256) void k_madd29(G1XYZZ* out, const G1Affine* in, int iters) {
    size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    Acc29 acc;
    acc.inf = true;
    G1Affine p = in[i + 1];
    xyzz_madd29(acc, in[i].x, in[i].y);
    for (int it = 0; it < iters; it++) { xyzz_madd29(acc, p.x, p.y); p.x.l[0] ^= acc.x.l[0] & 0xff; }
    out[i] = acc29_to_xyzz(acc);
}
__global__ void k_mul29(Fp* out, const Fp* in, int iters) {
    size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    U29 a = u29_load(in[i]), b = u29_load(in[i + 1]);
    for (int it = 0; it < iters; it++) { a = u29_mul(a, b); b = u29_mul(b, a); }
    out[i] = u29_store(u29_mul(a, b));
}

template <class K, class... A>
static double timeit(int blocks, int threads, K k, A... args) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, args...);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, args...);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s CUs=%d clock=%d kHz\n", prop.name, cus, prop.clockRate);
    uint64_t* out;
    CHECK(hipMalloc(&out, 64ull << 20));
    const int iters = 2000;
    // waves per SIMD sweep: blocks of 256 threads = 4 waves = 1 wave/SIMD per block per CU
    for (int wps : {1, 2, 4, 8}) {
        int blocks = cus * wps;
        double lanes = (double)blocks * 256;
#define RUN(name, kern, ilp, ...)                                                                                     \
    {                                                                                                                 \
        double ms = timeit(blocks, 256, kern<ilp>, out, __VA_ARGS__, iters);                                          \
        double ops = lanes * iters * 16.0 * ilp;                                                                      \
        double cyc_per_wave_instr = (ms * 1e-3 * 2.4e9) / (iters * 16.0 * ilp * wps);                                 \
        printf("%-14s ilp=%d waves/SIMD=%d  %8.3f ms  %8.2f Tops/s  ~%5.2f cyc/wave-instr/SIMD (at 2.4GHz)\n", name, ilp, wps, ms, ops / ms * 1e-9, cyc_per_wave_instr); \
    }
        RUN("mad_u64_u32", k_mad64, 1, 12345u, 67891u)
        RUN("mad_u64_u32", k_mad64, 4, 12345u, 67891u)
        RUN("mul_lo_u32", k_mullo, 1, 12345u, 67891u)
        RUN("mul_lo_u32", k_mullo, 4, 12345u, 67891u)
        RUN("addc_u32", k_add32, 1, 12345u, 67891u)
        RUN("addc_u32", k_add32, 4, 12345u, 67891u)
        RUN("fma_f64", k_fma64, 1, 1.000001, 0.5)
        RUN("fma_f64", k_fma64, 4, 1.000001, 0.5)
        RUN("lshl_add_u64", k_lshladd64, 4, (uint64_t)77)
        RUN("pk_mov_b32", k_pkmov, 4, (uint64_t)77)
    }
    // modmul throughput
    Fp* in;
    size_t nmax = (size_t)cus * 8 * 256 + 64;
    CHECK(hipMalloc(&in, nmax * sizeof(G1Affine)));
    std::vector<uint32_t> h(nmax * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) & ((i % 8 == 7) ? 0x0fffffffu : 0xffffffffu);
    CHECK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int wps : {1, 2, 4, 8}) {
        int blocks = cus * wps, it2 = 500;
        double ms = timeit(blocks, 256, k_modmul, (Fp*)out, (const Fp*)in, it2);
        double muls = (double)blocks * 256 * it2 * 2;
        printf("modmul  waves/SIMD=%d  %8.3f ms  %8.2f Gmul/s   %7.1f cyc/modmul/wave/SIMD\n", wps, ms, muls / ms * 1e-6, ms * 1e-3 * 2.4e9 / (it2 * 2.0 * wps));
        ms = timeit(blocks, 256, k_modadd, (Fp*)out, (const Fp*)in, it2);
        printf("modadd+sub waves/SIMD=%d  %8.3f ms  %8.2f Gop/s   %7.1f cyc/op/wave/SIMD\n", wps, ms, muls / ms * 1e-6, ms * 1e-3 * 2.4e9 / (it2 * 2.0 * wps));
    }
    for (int wps : {1, 2, 4}) {
        int blocks = cus * wps, it2 = 500;
        double ms = timeit(blocks, 256, k_mul29, (Fp*)out, (const Fp*)in, it2);
        double muls = (double)blocks * 256 * it2 * 2;
        printf("mul29   waves/SIMD=%d  %8.3f ms  %8.2f Gmul/s   %7.1f cyc/modmul/wave/SIMD\n", wps, ms, muls / ms * 1e-6, ms * 1e-3 * 2.4e9 / (it2 * 2.0 * wps));
        it2 = 200;
        ms = timeit(blocks, 256, k_madd29, (G1XYZZ*)out, (const G1Affine*)in, it2);
        printf("g1 madd29 waves/SIMD=%d  %8.3f ms  %8.2f Gadd/s   %7.1f cyc/madd/wave/SIMD\n", wps, ms, (double)blocks * 256 * it2 / ms * 1e-6, ms * 1e-3 * 2.4e9 / (it2 * 1.0 * wps));
    }
    for (int wps : {1, 2, 4}) {
        int blocks = cus * wps, it2 = 200;
        double ms = timeit(blocks, 256, k_madd, (G1XYZZ*)out, (const G1Affine*)in, it2);
        double adds = (double)blocks * 256 * it2;
        printf("g1 madd waves/SIMD=%d  %8.3f ms  %8.2f Gadd/s   %7.1f cyc/madd/wave/SIMD\n", wps, ms, adds / ms * 1e-6, ms * 1e-3 * 2.4e9 / (it2 * 1.0 * wps));
    }
    return 0;
}
