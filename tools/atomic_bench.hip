// micro-benchmark: throughput of global atomics for a counting sort of ~13.6 M (key, value) pairs over 2^19 counters
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void k_hist(uint32_t* cnt, uint32_t n, uint32_t mask) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&cnt[mix(i) & mask], 1u);
}
__global__ void k_scatter(uint32_t* cur, uint32_t* out, uint32_t n, uint32_t mask) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { uint32_t p = atomicAdd(&cur[mix(i) & mask], 1u); out[(p + (mix(i) & mask) * 26u) % n] = i; }
}
int main() {
    const uint32_t n = 13631488, nb = 1 << 19;
    uint32_t *cnt, *out; hipMalloc(&cnt, nb * 4); hipMalloc(&out, (size_t)n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(cnt, 0, nb * 4);
        hipEventRecord(e0); hipLaunchKernelGGL(k_hist, dim3((n + 255) / 256), dim3(256), 0, 0, cnt, n, nb - 1); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); printf("hist    %.3f ms (%.1f G atomics/s)\n", ms, n / ms / 1e6);
        hipMemset(cnt, 0, nb * 4);
        hipEventRecord(e0); hipLaunchKernelGGL(k_scatter, dim3((n + 255) / 256), dim3(256), 0, 0, cnt, out, n, nb - 1); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("scatter %.3f ms\n", ms);
    }
    return 0;
}
