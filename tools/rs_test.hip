// Stand-alone check and timing of the hand-written radix sort (csrc/radix.hpp) against std::stable_sort, outside the library:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Inoir_backend_using_gnark_amd/csrc tools/rs_test.hip -o tools/rs_test && tools/rs_test
// Sizes include the MSM's own (13.6 M pairs, 19-bit keys), ragged tails, every pass count (1..3), constant keys and a heavy skew (half the keys equal).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <numeric>
#include <vector>

#define ZKMI_PRIO_HI 3
#include "radix.hpp"
using namespace zkmi;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static uint64_t sm(uint64_t& s) { uint64_t z = (s += 0x9e3779b97f4a7c15ULL); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); }

static double run(size_t n, unsigned key_bits, int kind, bool check) {
    std::vector<uint32_t> k(n), v(n);
    uint64_t s = 0x1234 + n * 31 + key_bits;
    for (size_t i = 0; i < n; i++) {
        uint32_t r = (uint32_t)sm(s) & ((1u << key_bits) - 1);
        if (kind == 1) r = 5 & ((1u << key_bits) - 1);            // constant
        if (kind == 2 && (sm(s) & 1)) r = (1u << key_bits) - 1;   // half the keys equal (a giant bucket)
        k[i] = r;
        v[i] = (uint32_t)i;
    }
    const RsPlan P = rs_plan(n, key_bits);
    uint32_t *dk[2], *dv[2], *tmp;
    for (int b = 0; b < 2; b++) { CHECK(hipMalloc(&dk[b], n * 4 + 16)); CHECK(hipMalloc(&dv[b], n * 4 + 16)); }
    CHECK(hipMalloc(&tmp, P.tmp_bytes));
    uint32_t* tile_hist = tmp;
    uint32_t* ghist = tile_hist + (size_t)RS_MAX_BINS * P.ntiles;
    uint32_t* gbase = ghist + RS_MAX_PASSES * RS_MAX_BINS;
    RsArgs A;
    A.npass = P.npass;
    for (unsigned p = 0; p < RS_MAX_PASSES; p++) { A.shift[p] = P.shift[p]; A.bits[p] = P.bits[p]; }
    const unsigned nt = (unsigned)P.ntiles;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    int cur = 0;
    for (int rep = 0; rep < (check ? 2 : 5); rep++) {
        CHECK(hipMemcpy(dk[0], k.data(), n * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dv[0], v.data(), n * 4, hipMemcpyHostToDevice));
        cur = 0;
        CHECK(hipEventRecord(e0, 0));
        CHECK(hipMemsetAsync(ghist, 0, RS_MAX_PASSES * RS_MAX_BINS * 4, 0));
        hipLaunchKernelGGL(k_rs_hist, dim3(nt), dim3(RS_THREADS), 0, 0, (const uint32_t*)dk[0], (uint32_t)n, A, nt, ghist, tile_hist, (const uint32_t*)nullptr);
        hipLaunchKernelGGL(k_rs_bases, dim3(1), dim3(RS_MAX_BINS), 0, 0, (const uint32_t*)ghist, gbase, P.npass);
        for (unsigned p = 0; p < P.npass; p++) {
            if (p) hipLaunchKernelGGL(k_rs_tile_hist, dim3(nt), dim3(RS_THREADS), 0, 0, (const uint32_t*)dk[cur], (uint32_t)n, P.shift[p], P.bits[p], nt, tile_hist, (const uint32_t*)nullptr);
            hipLaunchKernelGGL(k_rs_scan_rows, dim3(1u << P.bits[p]), dim3(256), 0, 0, tile_hist, nt);
            hipLaunchKernelGGL(k_rs_scatter, dim3(nt), dim3(RS_THREADS), 0, 0, (const uint32_t*)dk[cur], (const uint32_t*)dv[cur], dk[cur ^ 1], dv[cur ^ 1], (uint32_t)n, P.shift[p],
                               P.bits[p], nt, (const uint32_t*)tile_hist, (const uint32_t*)(gbase + p * RS_MAX_BINS), (const uint32_t*)nullptr);
            cur ^= 1;
        }
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    int bad = 0;
    if (check) {
        std::vector<uint32_t> gk(n), gv(n), idx(n);
        CHECK(hipMemcpy(gk.data(), dk[cur], n * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(gv.data(), dv[cur], n * 4, hipMemcpyDeviceToHost));
        std::iota(idx.begin(), idx.end(), 0u);
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return k[a] < k[b]; });
        for (size_t i = 0; i < n && bad < 5; i++)
            if (gk[i] != k[idx[i]] || gv[i] != v[idx[i]]) { printf("  MISMATCH at %zu: got (%u, %u) want (%u, %u)\n", i, gk[i], gv[i], k[idx[i]], v[idx[i]]); bad++; }
    }
    if (!check || bad || n > 1000000) printf("n=%zu key_bits=%u kind=%d passes=%u: %.3f ms%s\n", n, key_bits, kind, P.npass, best, check ? (bad ? "  FAILED" : "  ok") : "");
    for (int b = 0; b < 2; b++) { CHECK(hipFree(dk[b])); CHECK(hipFree(dv[b])); }
    CHECK(hipFree(tmp));
    return bad ? -1 : best;
}

int main() {
    int fails = 0, cases = 0;
    const size_t sizes[] = {1, 63, 64, 65, 1000, 8191, 8192, 8193, 100000, (1 << 20) + 17};
    for (size_t n : sizes)
        for (unsigned kb : {1u, 5u, 8u, 9u, 13u, 16u, 17u, 19u, 22u})
            for (int kind : {0, 1, 2}) {
                cases++;
                if (run(n, kb, kind, true) < 0) fails++;
            }
    cases++;
    if (run(13631488, 19, 0, true) < 0) fails++;
    run(13631488, 19, 0, false);
    run(13631488, 19, 2, false);
    run(54525952, 21, 0, false);
    run((size_t)12 << 26, 21, 0, false);
    if (fails) printf("FAILED: %d of %d cases\n", fails, cases);
    else printf("all %d cases ok\n", cases);
    return fails ? 1 : 0;
}
