// Measurement tool (CPU only): the RawR1CS reader (csrc/acir_host.hpp raw_r1cs_build) on one payload -- the gates array split among the hardware threads against one
// reader, circuit only (what a resident circuit costs to make) against circuit + host-side wire vector.
//   g++ -O2 -std=c++17 tools/raw_lower_bench.cpp -lpthread -o /tmp/raw_lower_bench ; /tmp/raw_lower_bench <raw.json>
#include <chrono>
#include <cstdio>
#include <string>

#include "../noir_backend_using_gnark_amd/csrc/acir_host.hpp"

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::string text;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    char buf[1 << 16];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, k);
    fclose(f);
    std::string err;
    double t0 = now_ms();
    zkmi::RawR1CSBuilt C;
    const int rc = zkmi::raw_r1cs_build(text.data(), text.size(), &C, &err, false);
    const double t_circuit = now_ms() - t0;
    t0 = now_ms();
    zkmi::RawR1CSBuilt B;
    const int rc2 = zkmi::raw_r1cs_build(text.data(), text.size(), &B, &err, true);
    const double t_full = now_ms() - t0;
    zkmi::acir_detail::parallel_cfg().min_bytes = (size_t)-1;
    t0 = now_ms();
    zkmi::RawR1CSBuilt C1;
    const int rc3 = zkmi::raw_r1cs_build(text.data(), text.size(), &C1, &err, false);
    const double t_circuit1 = now_ms() - t0;
    bool same = rc == rc3 && rc2 == rc && C1.order == C.order && C1.prod_a == C.prod_a && C1.prod_b == C.prod_b;
    for (int m = 0; m < 3; m++)
        same = same && C1.ptr[m] == C.ptr[m] && C1.idx[m] == C.idx[m] && C1.val[m].size() == C.val[m].size() && !memcmp(C1.val[m].data(), C.val[m].data(), 32 * C.val[m].size()) &&
               B.idx[m] == C.idx[m];
    printf("{\"text_bytes\": %zu, \"constraints\": %zu, \"wires\": %zu, \"rc\": %d, \"circuit_only_ms\": %.1f, \"circuit_and_host_wires_ms\": %.1f, \"one_reader_circuit_only_ms\": %.1f, "
           "\"host_threads\": %u, \"same_output\": %s}\n",
           text.size(), C.ptr[0].size() - 1, 1 + C.n_values + C.prod_a.size(), rc, t_circuit, t_full, t_circuit1, std::thread::hardware_concurrency(), same ? "true" : "false");
    return same ? 0 : 1;
}
