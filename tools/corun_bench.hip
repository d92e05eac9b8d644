// What slows a short kernel down next to an accumulate kernel?  (neither wave slots nor issue priority.)  A controlled co-run:
// stream A runs a mixed-addition kernel at the accumulate kernel's occupancy (4 x 256-thread workgroups per CU, 125 VGPRs) in three flavours --
//     compute   no memory traffic at all: every lane adds the same register-resident point again and again
//     gather    before every mixed addition the lane reads a 64-byte record at a hashed index of a 6 GiB table (k_accumulate<G1>'s access pattern and rate)
//     gather2x  two such reads per mixed addition (twice the random traffic, same arithmetic)
// -- while stream B runs the hand-written radix sort (radix.hpp: 13.6 M pairs, 19-bit keys, 0.29 ms alone) back to back.  Reported: the sort's time alone and
// under each flavour, the mixed-addition kernel's time alone and under the sort.  If `compute` already slows the sort, the contention is on the CU
// (issue, LDS, wave slots); if only `gather` does, it is the memory system.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I noir_backend_using_gnark_amd/csrc tools/corun_bench.hip -o tools/corun_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "ff.hpp"
#include "curve.hpp"
#include "ff29.hpp"
#ifndef ZKMI_PRIO_HI
#define ZKMI_PRIO_HI 0  // build a second binary with -DZKMI_PRIO_HI=3: the sort kernels then raise their wave priority
#endif
#include "radix.hpp"
using namespace zkmi;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef Affine<Fp> G1A;
typedef XYZZ<Fp> G1X;

template <int GATHERS>
__global__ __launch_bounds__(256) void k_madd_load(G1X* out, const G1A* in, const uint4* table, unsigned log_rec, int iters) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Acc29 acc;
    acc.inf = true;
    G1A p = in[i & 4095];
    xyzz_madd29(acc, p.x, p.y);
    size_t h = i * 0x9E3779B97F4A7C15ULL;
    const size_t mask = ((size_t)1 << log_rec) - 1;
    for (int it = 0; it < iters; it++) {
        if (GATHERS) {
            uint32_t x = 0;
#pragma unroll
            for (int g = 0; g < GATHERS; g++) {
                h = h * 0xD6E8FEB86659FD93ULL + 0x632BE59BD9B4E019ULL;
                const uint4* r = table + ((h >> 20) & mask) * 4;
                const uint4 a = r[0], b = r[1], c = r[2], d = r[3];
                x ^= a.x ^ b.y ^ c.z ^ d.w;
            }
            p.x.l[0] ^= x & 0xff;  // the record feeds the addition (fill pattern: no effect on the value range)
        }
        xyzz_madd29(acc, p.x, p.y);
        p.x.l[1] ^= acc.x.l[0] & 0xff;
    }
    out[i] = acc29_to_xyzz(acc);
}

struct Sort {
    size_t n;
    unsigned key_bits;
    RsPlan P;
    uint32_t *k[2], *v[2], *tmp;
};
static void sort_enqueue(const Sort& S, hipStream_t st) {
    uint32_t* tile_hist = S.tmp;
    uint32_t* ghist = tile_hist + (size_t)RS_MAX_BINS * S.P.ntiles;
    uint32_t* gbase = ghist + RS_MAX_PASSES * RS_MAX_BINS;
    RsArgs A;
    A.npass = S.P.npass;
    for (unsigned p = 0; p < RS_MAX_PASSES; p++) { A.shift[p] = S.P.shift[p]; A.bits[p] = S.P.bits[p]; }
    const unsigned nt = (unsigned)S.P.ntiles;
    int cur = 0;
    CHECK(hipMemsetAsync(ghist, 0, RS_MAX_PASSES * RS_MAX_BINS * 4, st));
    hipLaunchKernelGGL(k_rs_hist, dim3(nt), dim3(RS_THREADS), 0, st, (const uint32_t*)S.k[0], (uint32_t)S.n, A, nt, ghist, tile_hist, (const uint32_t*)nullptr);
    hipLaunchKernelGGL(k_rs_bases, dim3(1), dim3(RS_MAX_BINS), 0, st, (const uint32_t*)ghist, gbase, S.P.npass);
    for (unsigned p = 0; p < S.P.npass; p++) {
        if (p) hipLaunchKernelGGL(k_rs_tile_hist, dim3(nt), dim3(RS_THREADS), 0, st, (const uint32_t*)S.k[cur], (uint32_t)S.n, S.P.shift[p], S.P.bits[p], nt, tile_hist, (const uint32_t*)nullptr);
        hipLaunchKernelGGL(k_rs_scan_rows, dim3(1u << S.P.bits[p]), dim3(256), 0, st, tile_hist, nt);
        hipLaunchKernelGGL(k_rs_scatter, dim3(nt), dim3(RS_THREADS), 0, st, (const uint32_t*)S.k[cur], (const uint32_t*)S.v[cur], S.k[cur ^ 1], S.v[cur ^ 1], (uint32_t)S.n, S.P.shift[p],
                           S.P.bits[p], nt, (const uint32_t*)tile_hist, (const uint32_t*)(gbase + p * RS_MAX_BINS), (const uint32_t*)nullptr);
        cur ^= 1;
    }
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipStream_t sa, sb;
    CHECK(hipStreamCreate(&sa));
    CHECK(hipStreamCreate(&sb));
    // the sort's data
    Sort S;
    S.n = 13631488;
    S.key_bits = 19;
    S.P = rs_plan(S.n, S.key_bits);
    for (int b = 0; b < 2; b++) { CHECK(hipMalloc(&S.k[b], S.n * 4)); CHECK(hipMalloc(&S.v[b], S.n * 4)); }
    CHECK(hipMalloc(&S.tmp, S.P.tmp_bytes));
    {
        std::vector<uint32_t> hk(S.n);
        uint64_t s = 1;
        for (size_t i = 0; i < S.n; i++) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; hk[i] = (uint32_t)(s >> 40) & ((1u << 19) - 1); }
        CHECK(hipMemcpy(S.k[0], hk.data(), S.n * 4, hipMemcpyHostToDevice));
        CHECK(hipMemset(S.v[0], 0, S.n * 4));
    }
    // the mixed-addition kernel's data: 4096 points (any field elements do for timing), a 6 GiB table of 64-byte records
    const unsigned log_rec = 26;  // 2^26 records x 64 B = 4 GiB (hashed index space); allocation 4 GiB
    uint4* table;
    CHECK(hipMalloc(&table, ((size_t)1 << log_rec) * 64));
    CHECK(hipMemset(table, 0, ((size_t)1 << log_rec) * 64));
    G1A* in;
    CHECK(hipMalloc(&in, 4096 * sizeof(G1A)));
    {
        std::vector<uint32_t> h(4096 * 16);
        for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) & ((i % 8 == 7) ? 0x0fffffffu : 0xffffffffu);
        CHECK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    G1X* out;
    CHECK(hipMalloc(&out, (size_t)cus * 4 * 256 * sizeof(G1X)));
    hipEvent_t e0, e1, f0, f1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&f0)); CHECK(hipEventCreate(&f1));
    auto launch_a = [&](int flavour, int blocks, int iters) {
        if (flavour == 0) hipLaunchKernelGGL((k_madd_load<0>), dim3(blocks), dim3(256), 0, sa, out, (const G1A*)in, (const uint4*)table, log_rec, iters);
        if (flavour == 1) hipLaunchKernelGGL((k_madd_load<1>), dim3(blocks), dim3(256), 0, sa, out, (const G1A*)in, (const uint4*)table, log_rec, iters);
        if (flavour == 2) hipLaunchKernelGGL((k_madd_load<2>), dim3(blocks), dim3(256), 0, sa, out, (const G1A*)in, (const uint4*)table, log_rec, iters);
    };
    const char* names[3] = {"compute", "gather", "gather2x"};
    const int NS = 8;  // sorts per measurement
    float sort_alone = 0;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(f0, sb));
        for (int k = 0; k < NS; k++) sort_enqueue(S, sb);
        CHECK(hipEventRecord(f1, sb));
        CHECK(hipEventSynchronize(f1));
        CHECK(hipEventElapsedTime(&sort_alone, f0, f1));
    }
    printf("{\"sort_wave_priority\": %d, \"eight_sorts_alone_ms\": %.3f, \"rows\": [\n", ZKMI_PRIO_HI, sort_alone);
    bool first = true;
    for (int wg : {4, 3, 2, 1})
        for (int fl = 0; fl < 2; fl++) {
            const int blocks = cus * wg, iters = 1200;  // ~20 ms at 4 workgroups per CU: the eight sorts finish well inside it
            float a_alone = 0, a_co = 0, s_co = 0;
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, sa));
            launch_a(fl, blocks, iters);
            CHECK(hipEventRecord(e1, sa));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&a_alone, e0, e1));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, sa));
            launch_a(fl, blocks, iters);
            CHECK(hipEventRecord(e1, sa));
            CHECK(hipEventRecord(f0, sb));
            for (int k = 0; k < NS; k++) sort_enqueue(S, sb);
            CHECK(hipEventRecord(f1, sb));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventSynchronize(f1));
            CHECK(hipEventElapsedTime(&a_co, e0, e1));
            CHECK(hipEventElapsedTime(&s_co, f0, f1));
            const double madds = (double)blocks * 256 * iters;
            printf("%s {\"madd_workgroups_per_cu\": %d, \"flavour\": \"%s\", \"madd_kernel_alone_ms\": %.3f, \"madd_per_s_alone\": %.3e, \"madd_kernel_with_sorts_ms\": %.3f, "
                   "\"eight_sorts_under_it_ms\": %.3f, \"sorts_inside_the_kernel\": %s, \"sort_slowdown\": %.2f, \"madd_kernel_slowdown_ms\": %.3f}",
                   first ? "" : ",\n", wg, names[fl], a_alone, madds / a_alone * 1e3, a_co, s_co, s_co < a_co ? "true" : "false", s_co / sort_alone, a_co - a_alone);
            first = false;
        }
    printf("\n]}\n");
    return 0;
}
