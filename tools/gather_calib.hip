// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access pattern of the MSM accumulate kernel: one 64-byte record per lane (four dwordx4 loads) at
// an unpredictable index of a table far larger than L2 + Infinity Cache.  MI355X_MICROARCH.md calibrates the counter only for wide coalesced streams
// ("FETCH_SIZE reports exactly 1/2 of the bytes ... other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//
//   hipcc --offload-arch=gfx950 -O3 tools/gather_calib.hip -o tools/gather_calib
//   rocprofv3 --pmc FETCH_SIZE -d <dir> -- tools/gather_calib          (then tools/gather_calib_summary.py <dir>)
//
// Kernels (each reads a KNOWN number of bytes, printed on stdout as "<kernel> bytes=<n>"):
//   k_stream16     every lane 16 B, fully coalesced, each byte once                  -- the guide's calibration case (expect FETCH_SIZE = bytes / 2)
//   k_gather64     every lane one 64-B record at a hashed index, each record once   -- k_accumulate<G1>'s table gathers
//   k_gather128    every lane one 128-B record at a hashed index, each record once  -- k_accumulate<G2>'s
//   k_gather64_seq every lane one 64-B record, consecutive lanes consecutive records -- 64-B accesses without the randomness
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_stream16(const uint4* __restrict__ src, size_t n16, uint4* __restrict__ sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = src[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;  // keeps the loads alive; never true for the fill pattern
}
// index of the record lane i reads: a bijection of [0, 2^log_n) (odd multiplier, xor-shift) -- every record exactly once, no locality
__device__ __forceinline__ size_t scramble(size_t i, unsigned log_n) {
    const size_t mask = ((size_t)1 << log_n) - 1;
    size_t x = (i * 0x9E3779B97F4A7C15ULL) & mask;
    x ^= x >> (log_n / 2);
    x = (x * 0xD6E8FEB86659FD93ULL) & mask;
    return x;
}
template <int REC16, bool RANDOM>
__global__ __launch_bounds__(256) void k_gather(const uint4* __restrict__ src, unsigned log_n, uint4* __restrict__ sink) {
    const size_t n = (size_t)1 << log_n;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = RANDOM ? scramble(i, log_n) : i;
        const uint4* p = src + r * REC16;
#pragma unroll
        for (int k = 0; k < REC16; k++) {
            uint4 v = p[k];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}

int main() {
    const unsigned log_rec = 26;                       // 2^26 records of 64 B = 4 GiB (128-B records: 2^25) >> 32 MiB L2 + 256 MiB Infinity Cache
    const size_t bytes = ((size_t)1 << log_rec) * 64;
    uint4 *d = nullptr, *sink = nullptr;
    CHECK(hipMalloc(&d, bytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(d, 0x5a, bytes));
    CHECK(hipDeviceSynchronize());
    const dim3 grid(256 * 8), block(256);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_stream16, grid, block, 0, 0, d, bytes / 16, sink);
        hipLaunchKernelGGL((k_gather<4, true>), grid, block, 0, 0, d, log_rec, sink);
        hipLaunchKernelGGL((k_gather<8, true>), grid, block, 0, 0, d, log_rec - 1, sink);
        hipLaunchKernelGGL((k_gather<4, false>), grid, block, 0, 0, d, log_rec, sink);
        CHECK(hipDeviceSynchronize());
    }
    printf("k_stream16 bytes=%zu\nk_gather<4, true> bytes=%zu\nk_gather<8, true> bytes=%zu\nk_gather<4, false> bytes=%zu\n", bytes, bytes, bytes, bytes);
    CHECK(hipFree(d));
    CHECK(hipFree(sink));
    return 0;
}
