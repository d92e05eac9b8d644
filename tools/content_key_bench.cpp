// Throughput of the content key (csrc/acir_host.hpp: a keyed SipHash-2-4-128 tree) on this host, vector and scalar forms, and its self-test.
//   g++ -O3 -std=c++17 -pthread -I include tools/content_key_bench.cpp -o /tmp/content_key_bench && /tmp/content_key_bench [MB]
#include <stdio.h>
#include <chrono>
#include "../noir_backend_using_gnark_amd/csrc/acir_host.hpp"
using namespace zkmi;
int main(int argc, char** argv) {
    const size_t n = (size_t)(argc > 1 ? atol(argv[1]) : 640) << 20;  // 0.63 GB: the two texts of a 2^20-constraint ProveWithPK call
    std::vector<char> buf(n);
    for (size_t i = 0; i < n; i++) buf[i] = "0123456789abcdef"[(i * 2654435761u >> 9) & 15];
    printf("{\"selftest\": %d, \"bytes\": %zu, \"threads\": %u", (int)siphash_selftest(), n, std::min(16u, std::thread::hardware_concurrency()));
    for (int scalar = 0; scalar < 2; scalar++) {
        sip_scalar_only() = scalar != 0;
        double best = 1e30;
        ContentKey K;
        for (int r = 0; r < 5; r++) {
            auto t0 = std::chrono::steady_clock::now();
            K = content_key(buf.data(), n);
            best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        printf(", \"%s_ms\": %.2f, \"%s_GBps\": %.1f", scalar ? "scalar" : "avx2", best, scalar ? "scalar" : "avx2", n / best / 1e6);
    }
    printf("}\n");
    return 0;
}
