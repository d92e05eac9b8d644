// Second round of instruction-rate probes (simple integer ops) for the unsaturated-limb field design.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define DEF32(NAME, ASM)                                                                   \
    __global__ void NAME(uint64_t* out, uint32_t a, uint32_t b, int iters) {               \
        uint32_t acc[4];                                                                   \
        for (int i = 0; i < 4; i++) acc[i] = threadIdx.x + i + a;                          \
        uint32_t x = b + threadIdx.x;                                                      \
        for (int it = 0; it < iters; it++) {                                               \
            _Pragma("unroll") for (int u = 0; u < 16; u++) {                               \
                _Pragma("unroll") for (int i = 0; i < 4; i++) asm volatile(ASM : "+v"(acc[i]) : "v"(x) : "vcc"); \
            }                                                                              \
        }                                                                                  \
        uint64_t s = 0;                                                                    \
        for (int i = 0; i < 4; i++) s += acc[i];                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                    \
    }
#define DEF64(NAME, ASM)                                                                   \
    __global__ void NAME(uint64_t* out, uint32_t a, uint32_t b, int iters) {               \
        uint64_t acc[4];                                                                   \
        for (int i = 0; i < 4; i++) acc[i] = threadIdx.x + i + a;                          \
        uint32_t x = b + threadIdx.x;                                                      \
        for (int it = 0; it < iters; it++) {                                               \
            _Pragma("unroll") for (int u = 0; u < 16; u++) {                               \
                _Pragma("unroll") for (int i = 0; i < 4; i++) asm volatile(ASM : "+v"(acc[i]) : "v"(x) : "vcc"); \
            }                                                                              \
        }                                                                                  \
        uint64_t s = 0;                                                                    \
        for (int i = 0; i < 4; i++) s += acc[i];                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                    \
    }
DEF32(k_add_u32, "v_add_u32_e32 %0, %0, %1")
DEF32(k_add_co, "v_add_co_u32_e32 %0, vcc, %0, %1")
DEF32(k_and, "v_and_b32_e32 %0, %0, %1")
DEF32(k_add3, "v_add3_u32 %0, %0, %1, %1")
DEF32(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
DEF32(k_and_or, "v_and_or_b32 %0, %0, %1, %1")
DEF32(k_bfe, "v_bfe_u32 %0, %0, 3, 29")
DEF32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 29")
DEF32(k_mul_u24, "v_mul_u32_u24_e32 %0, %0, %1")
DEF32(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %0")
DEF32(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
DEF32(k_cndmask, "v_cndmask_b32_e32 %0, %0, %1, vcc")
DEF32(k_mov, "v_mov_b32_e32 %0, %1")
DEF32(k_fma32, "v_fma_f32 %0, %0, %1, %1")
DEF32(k_sub_co, "v_subb_co_u32_e32 %0, vcc, %0, %1, vcc")
DEF64(k_lshr64, "v_lshrrev_b64 %0, 29, %0")
DEF64(k_mad64_lit0, "v_mad_u64_u32 %0, vcc, %1, %1, %0")
DEF64(k_mad64_s, "v_mad_u64_u32 %0, s[10:11], %1, %1, %0")

template <class K>
static double timeit(int blocks, K k, uint64_t* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u, iters);
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
    uint64_t* out;
    CHECK(hipMalloc(&out, 64ull << 20));
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {
        int blocks = cus * wps;
#define RUN(name, kern) { double ms = timeit(blocks, kern, out, iters); printf("%-14s waves/SIMD=%d %8.3f ms  ~%5.2f cyc/wave-instr/SIMD\n", name, wps, ms, (ms * 1e-3 * 2.4e9) / (iters * 64.0 * wps)); }
        RUN("add_u32", k_add_u32) RUN("add_co_u32", k_add_co) RUN("subb_co", k_sub_co) RUN("and_b32", k_and) RUN("add3_u32", k_add3)
        RUN("lshl_add_u32", k_lshl_add) RUN("and_or_b32", k_and_or) RUN("bfe_u32", k_bfe) RUN("alignbit", k_alignbit)
        RUN("mul_u32_u24", k_mul_u24) RUN("mad_u32_u24", k_mad_u24) RUN("mul_hi_u32", k_mul_hi) RUN("cndmask", k_cndmask)
        RUN("mov_b32", k_mov) RUN("fma_f32", k_fma32) RUN("lshrrev_b64", k_lshr64) RUN("mad64 vcc", k_mad64_lit0) RUN("mad64 sdst", k_mad64_s)
    }
    return 0;
}
