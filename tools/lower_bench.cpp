// Measurement tool (CPU only): the streaming ACIR front end (csrc/acir_host.hpp) against the document-tree reader it replaced (tests/cpp/json_dom_ref.hpp)
// on one ACIR text.   g++ -O2 -std=c++17 tools/lower_bench.cpp -lpthread -o /tmp/lower_bench ; /tmp/lower_bench <acir.json> <n_values> [skip_tree]
#include <chrono>
#include <cstdio>
#include <string>

#include "../noir_backend_using_gnark_amd/csrc/acir_host.hpp"
#include "../tests/cpp/json_dom_ref.hpp"

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static bool slurp(const char* path, std::string* text) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) text->append(buf, k);
    fclose(f);
    return true;
}

int main(int argc, char** argv) {
    if (argc >= 3 && !strcmp(argv[1], "--keys")) {  // the content keys of several texts from ONE process (the key's masks are drawn per process)
        printf("[");
        for (int i = 2; i < argc; i++) {
            std::string t;
            if (!slurp(argv[i], &t)) return 2;
            const zkmi::ContentKey k = zkmi::content_key(t.data(), t.size());
            printf("%s\"%016llx%016llx\"", i > 2 ? ", " : "", (unsigned long long)k.h[0], (unsigned long long)k.h[1]);
        }
        printf("]\n");
        return 0;
    }
    if (argc < 3) return 2;
    std::string text;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    char buf[1 << 16];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, k);
    fclose(f);
    const size_t n_values = strtoull(argv[2], nullptr, 10);
    double t0 = now_ms();
    const zkmi::ContentKey ck = zkmi::content_key(text.data(), text.size());
    double t_key = now_ms() - t0;
    zkmi::Gates W, G;
    std::string err;
    t0 = now_ms();
    int rc = zkmi::lower_acir(text.data(), text.size(), n_values, ZK_ACIR_LAYOUT_REFERENCE, false, &W, &err);
    double t_wiring = now_ms() - t0;
    t0 = now_ms();
    int rc2 = zkmi::lower_acir(text.data(), text.size(), n_values, ZK_ACIR_LAYOUT_REFERENCE, true, &G, &err);
    double t_full = now_ms() - t0;
    // the same two runs on ONE thread (the opcodes array is split among the hardware threads by default: acir_detail::elements_parallel)
    zkmi::acir_detail::parallel_cfg().min_bytes = (size_t)-1;
    zkmi::Gates W1, G1;
    t0 = now_ms();
    const int rc3 = zkmi::lower_acir(text.data(), text.size(), n_values, ZK_ACIR_LAYOUT_REFERENCE, false, &W1, &err);
    const double t_wiring1 = now_ms() - t0;
    t0 = now_ms();
    const int rc4 = zkmi::lower_acir(text.data(), text.size(), n_values, ZK_ACIR_LAYOUT_REFERENCE, true, &G1, &err);
    const double t_full1 = now_ms() - t0;
    double t_tree = -1;
    bool same = rc3 == rc && rc4 == rc2 && W1.xa == W.xa && W1.xb == W.xb && W1.xc == W.xc && W1.order == W.order && G1.xa == G.xa && G1.xc == G.xc && G1.ql.size() == G.ql.size() &&
                !memcmp(G1.ql.data(), G.ql.data(), 32 * G.ql.size()) && !memcmp(G1.qk.data(), G.qk.data(), 32 * G.qk.size()) && !memcmp(G1.qo.data(), G.qo.data(), 32 * G.qo.size());
    if (argc < 4) {
        domref::Gates D;
        t0 = now_ms();
        int rd = domref::lower_acir(text.data(), text.size(), n_values, ZK_ACIR_LAYOUT_REFERENCE, &D);
        t_tree = now_ms() - t0;
        same = same && rd == rc2 && D.xa == G.xa && D.xb == G.xb && D.xc == G.xc && D.order == G.order && D.ql.size() == G.ql.size() &&
               !memcmp(D.ql.data(), G.ql.data(), 32 * G.ql.size()) && !memcmp(D.qk.data(), G.qk.data(), 32 * G.qk.size()) && !memcmp(D.qm.data(), G.qm.data(), 32 * G.qm.size());
    }
    printf("{\"text_bytes\": %zu, \"gates\": %zu, \"n_vars\": %zu, \"n_public\": %zu, \"rc\": [%d, %d], \"content_key_ms\": %.2f, \"streaming_wiring_ms\": %.1f, "
           "\"streaming_with_coefficients_ms\": %.1f, \"one_thread_wiring_ms\": %.1f, \"one_thread_with_coefficients_ms\": %.1f, \"host_threads\": %u, \"document_tree_ms\": %.1f, "
           "\"same_output\": %s, \"key\": \"%016llx%016llx\"}\n",
           text.size(), G.xa.size(), G.n_vars, G.n_public, rc, rc2, t_key, t_wiring, t_full, t_wiring1, t_full1, std::thread::hardware_concurrency(), t_tree, same ? "true" : "false",
           (unsigned long long)ck.h[0], (unsigned long long)ck.h[1]);
    return same ? 0 : 1;
}
