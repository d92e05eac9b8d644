// Micro-benchmark: the scalar-preparation sort of an MSM (u32 bucket keys of `bits` bits, u32 payload) through rocPRIM's onesweep with other
// radix widths / tile shapes than the library default (8 bits per pass, 3 passes for the 19 key bits of the c = 20 window tables).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 sort_bench.hip -o sort_bench ; run: ./sort_bench [n=13631488] [bits=19]
#include <hip/hip_runtime.h>
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

__global__ void k_fill(uint32_t* k, uint32_t* v, size_t n, uint32_t mask) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    k[i] = (uint32_t)(z >> 33) & mask;
    v[i] = (uint32_t)i;
}
__global__ void k_check(const uint32_t* k, size_t n, int* bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 < n && k[i] > k[i + 1]) atomicAdd(bad, 1);
}

template <class Config>
static void run(const char* name, size_t n, unsigned bits, uint32_t* k0, uint32_t* k1, uint32_t* v0, uint32_t* v1, uint32_t* ksrc, uint32_t* vsrc, int* d_bad) {
    size_t tmp_bytes = 0;
    rocprim::double_buffer<uint32_t> kb(k0, k1), vb(v0, v1);
    CK(rocprim::radix_sort_pairs<Config>(nullptr, tmp_bytes, kb, vb, n, 0, bits, 0));
    void* tmp;
    CK(hipMalloc(&tmp, tmp_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0;
    const int reps = 10;
    for (int r = 0; r < reps + 2; r++) {
        CK(hipMemcpyAsync(k0, ksrc, n * 4, hipMemcpyDeviceToDevice, 0));
        CK(hipMemcpyAsync(v0, vsrc, n * 4, hipMemcpyDeviceToDevice, 0));
        rocprim::double_buffer<uint32_t> kb2(k0, k1), vb2(v0, v1);
        CK(hipEventRecord(e0, 0));
        CK(rocprim::radix_sort_pairs<Config>(tmp, tmp_bytes, kb2, vb2, n, 0, bits, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) { sum += ms; if (ms < best) best = ms; }
        if (r == reps + 1) {
            CK(hipMemset(d_bad, 0, 4));
            k_check<<<(unsigned)((n + 255) / 256), 256>>>(kb2.current(), n, d_bad);
            int bad;
            CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
            std::printf("{\"config\": \"%s\", \"n\": %zu, \"bits\": %u, \"avg_ms\": %.4f, \"min_ms\": %.4f, \"GBps_16B_per_item\": %.1f, \"tmp_MB\": %.1f, \"sorted\": %s}\n", name, n, bits, sum / reps,
                        best, 16.0 * n / (sum / reps * 1e-3) / 1e9, tmp_bytes / 1048576.0, bad ? "false" : "true");
        }
    }
    CK(hipFree(tmp));
}

template <unsigned RB, unsigned BS, unsigned IPT>
using OS = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                      rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 12>, rocprim::kernel_config<BS, IPT>, RB, rocprim::block_radix_rank_algorithm::match>>;

int main(int argc, char** argv) {
    size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 13631488;
    unsigned bits = argc > 2 ? atoi(argv[2]) : 19;
    uint32_t *k0, *k1, *v0, *v1, *ks, *vs;
    int* d_bad;
    for (uint32_t** p : {&k0, &k1, &v0, &v1, &ks, &vs}) CK(hipMalloc(p, n * 4));
    CK(hipMalloc(&d_bad, 4));
    k_fill<<<(unsigned)((n + 255) / 256), 256>>>(ks, vs, n, (1u << bits) - 1);
    CK(hipDeviceSynchronize());
#define RUN(name, ...) run<__VA_ARGS__>(name, n, bits, k0, k1, v0, v1, ks, vs, d_bad)
    RUN("default", rocprim::default_config);
#ifndef SKIP_CUSTOM
    RUN("rb8_bs512_ipt12", OS<8, 512, 12>);
    RUN("rb8_bs1024_ipt8", OS<8, 1024, 8>);
    RUN("rb7_bs512_ipt12", OS<7, 512, 12>);
    RUN("rb10_bs512_ipt12", OS<10, 512, 12>);
    RUN("rb10_bs256_ipt16", OS<10, 256, 16>);
    RUN("rb10_bs1024_ipt8", OS<10, 1024, 8>);
    RUN("rb10_bs512_ipt20", OS<10, 512, 20>);
    RUN("rb9_bs512_ipt12", OS<9, 512, 12>);
#endif
    return 0;
}
