// TEST INFRASTRUCTURE: deterministic mutation run over the host-side parsers of untrusted text (CPU only; built by tests/cpp/Makefile with
// -fsanitize=address,undefined).  The reference ends the process cleanly on any parse failure (gnark_backend_ffi/main.go:26-30,46-50,61-72: every
// error is a log.Fatal) -- here that means: every mutant is either accepted or rejected with a status code, never undefined behaviour, and the
// streaming front end (csrc/acir_host.hpp) agrees with the document-tree reader it replaced (tests/cpp/json_dom_ref.hpp) on accept / reject, on
// the status code and on every output word.
//   parser_fuzz <cases> <seed file>...      seed files: *.acir.json (ACIR), *.raw.json (RawR1CS), *.felts.hex (felt vector), *.bin (key images)
// Mutations: bit flips, byte overwrites from a dictionary of structural characters, deletions, insertions, truncations, slice duplications,
// digit-run edits (length fields / indices), escapes spliced into strings.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../noir_backend_using_gnark_amd/csrc/acir_host.hpp"
#include "../../noir_backend_using_gnark_amd/csrc/text_host.hpp"
#include "json_dom_ref.hpp"

static uint64_t g_state = 0x9e3779b97f4a7c15ULL;
static uint64_t rnd() {  // SplitMix64
    uint64_t z = (g_state += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static size_t below(size_t n) { return n ? (size_t)(rnd() % n) : 0; }

static std::string mutate(const std::string& seed) {
    static const char* dict[] = {"{", "}", "[", "]", "\"", ",", ":", "\\", "\\u0041", "\\\"", "0", "9", "-", ".", "e", "1e2", "4294967296", "4294967295", "null", "true",
                                 "false", "Arithmetic", "Directive", "mul_terms", "linear_combinations", "q_c", "opcodes", "public_inputs", " ", "\n", "\f", "nan", "0x10",
                                 "\"Arithmetic\":", "\"q_c\":\"01\",", "[[\"01\",1,2]]", "gates", "values", "coefficient", "sum", "multiplier", "multiplicand", "g", "ff"};
    std::string s = seed;
    const int rounds = 1 + (int)below(4);
    for (int r = 0; r < rounds; r++) {
        const size_t n = s.size();
        switch (below(9)) {
            case 0: if (n) s[below(n)] ^= (char)(1u << below(8)); break;
            case 1: if (n) { const char* d = dict[below(sizeof dict / sizeof *dict)]; s.replace(below(n), below(3), d); } break;
            case 2: if (n) s.erase(below(n), 1 + below(8)); break;
            case 3: s.insert(below(n + 1), dict[below(sizeof dict / sizeof *dict)]); break;
            case 4: if (n) s.resize(below(n)); break;
            case 5: if (n) { const size_t a = below(n), l = 1 + below(n - a < 200 ? n - a : 200); s.insert(below(n + 1), s.substr(a, l)); } break;
            case 6: {  // edit a run of digits (indices, counts)
                size_t a = below(n + 1);
                while (a < n && !(s[a] >= '0' && s[a] <= '9')) a++;
                if (a < n) { size_t b = a; while (b < n && s[b] >= '0' && s[b] <= '9' && b - a < 12) b++; s.replace(a, b - a, std::to_string(rnd() >> below(64))); }
                break;
            }
            case 7: if (n) { const size_t a = below(n); s.insert(a, std::string(1 + below(70), "[{"[below(2)])); } break;  // nesting
            case 8: if (n) { const size_t a = below(n), b = below(n); std::swap(s[a], s[b]); } break;
        }
    }
    return s;
}

static int failures = 0;
#define CHECK(cond, ...)                                        \
    do {                                                        \
        if (!(cond)) {                                          \
            if (failures++ < 20) { fprintf(stderr, "MISMATCH: " __VA_ARGS__); fprintf(stderr, "\n"); } \
        }                                                       \
    } while (0)

template <class VA, class VB>
static bool same_frs(const VA& a, const VB& b) { return a.size() == b.size() && (a.empty() || !memcmp(a.data(), b.data(), a.size() * 32)); }

static void check_acir(const std::string& text, size_t n_values, int layout, const char* what, size_t id) {
    zkmi::Gates G, W;
    std::string err, err2;
    const int rc = zkmi::lower_acir(text.data(), text.size(), n_values, layout, true, &G, &err);
    const int rcw = zkmi::lower_acir(text.data(), text.size(), n_values, layout, false, &W, &err2);
    domref::Gates D;
    const int rd = domref::lower_acir(text.data(), text.size(), n_values, layout, &D);
    CHECK(rc == rd, "%s #%zu (n_values %zu, layout %d): streaming rc %d (%s) vs tree rc %d (%s)", what, id, n_values, layout, rc, err.c_str(), rd, domref::g_err.c_str());
    CHECK(rc == rcw, "%s #%zu: rc with coefficients %d, wiring only %d", what, id, rc, rcw);
    if ((id & 3) == 0) {  // (every fourth case: a thread start costs more than these short parses) the same text with the opcodes array split among 2..5 parsers (acir_detail::elements_parallel; forced on these short texts): status, message and
        // every output word of the one-thread run
        zkmi::acir_detail::ParallelCfg& pc = zkmi::acir_detail::parallel_cfg();
        const zkmi::acir_detail::ParallelCfg keep = pc;
        pc.min_bytes = 0;
        pc.threads = 2 + (unsigned)below(4);
        zkmi::Gates GP, WP;
        std::string ep, ep2;
        const int rp = zkmi::lower_acir(text.data(), text.size(), n_values, layout, true, &GP, &ep);
        const int rpw = zkmi::lower_acir(text.data(), text.size(), n_values, layout, false, &WP, &ep2);
        pc = keep;
        CHECK(rp == rc && rpw == rcw && ep == err && ep2 == err2, "%s #%zu: %u parsers rc %d (%s) / %d (%s) vs one parser rc %d (%s) / %d (%s)", what, id, pc.threads, rp, ep.c_str(), rpw,
              ep2.c_str(), rc, err.c_str(), rcw, err2.c_str());
        if (rc == ZK_OK && rp == ZK_OK) {
            CHECK(GP.xa == G.xa && GP.xb == G.xb && GP.xc == G.xc && GP.order == G.order && GP.n_public == G.n_public && GP.n_vars == G.n_vars, "%s #%zu: parallel wiring", what, id);
            CHECK(same_frs(GP.ql, G.ql) && same_frs(GP.qr, G.qr) && same_frs(GP.qo, G.qo) && same_frs(GP.qm, G.qm) && same_frs(GP.qk, G.qk), "%s #%zu: parallel coefficients", what, id);
        }
        if (rcw == ZK_OK && rpw == ZK_OK) CHECK(WP.xa == W.xa && WP.xb == W.xb && WP.xc == W.xc && WP.order == W.order, "%s #%zu: parallel wiring-only mode", what, id);
    }
    if (rc != ZK_OK || rd != ZK_OK) return;
    CHECK(G.n_public == D.n_public && G.n_vars == D.n_vars, "%s #%zu: n_public / n_vars", what, id);
    CHECK(G.xa == D.xa && G.xb == D.xb && G.xc == D.xc && G.order == D.order, "%s #%zu: wiring / order", what, id);
    CHECK(same_frs(G.ql, D.ql) && same_frs(G.qr, D.qr) && same_frs(G.qo, D.qo) && same_frs(G.qm, D.qm) && same_frs(G.qk, D.qk), "%s #%zu: coefficients", what, id);
    CHECK(W.xa == D.xa && W.xb == D.xb && W.xc == D.xc && W.order == D.order && W.n_public == D.n_public, "%s #%zu: wiring-only mode", what, id);
}

template <class A, class B>
static bool same_words(const A& a, const B& b) { return a.size() == b.size() && std::equal(a.begin(), a.end(), b.begin()); }
static void check_raw(const std::string& text, const char* what, size_t id) {
    zkmi::RawR1CSBuilt B;
    std::string err;
    const int rc = zkmi::raw_r1cs_build(text.data(), text.size(), &B, &err);
    domref::RawBuilt D;
    const int rd = domref::raw_r1cs_build(text.data(), text.size(), &D);
    CHECK(rc == rd, "%s #%zu: streaming rc %d (%s) vs tree rc %d (%s)", what, id, rc, err.c_str(), rd, domref::g_err.c_str());
    if ((id & 3) == 0) {  // the gates array split among 2..5 readers (raw_detail::elements_parallel, forced on these short texts): status, message, every output word
        zkmi::RawR1CSBuilt P;
        std::string ep;
        zkmi::acir_detail::ParallelCfg& pc = zkmi::acir_detail::parallel_cfg();
        const zkmi::acir_detail::ParallelCfg keep = pc;
        pc.min_bytes = 0;
        pc.threads = 2 + (unsigned)below(4);
        const int rp = zkmi::raw_r1cs_build(text.data(), text.size(), &P, &ep);
        pc = keep;
        CHECK(rp == rc && ep == err, "%s #%zu: parallel gate readers rc %d (%s) vs one reader rc %d (%s)", what, id, rp, ep.c_str(), rc, err.c_str());
        if (rp == ZK_OK && rc == ZK_OK) {
            bool same = P.n_public == B.n_public && same_frs(P.wires, B.wires) && P.order == B.order && P.prod_a == B.prod_a && P.prod_b == B.prod_b && P.values_at == B.values_at &&
                        P.values_len == B.values_len;
            for (int m = 0; m < 3; m++) same = same && P.ptr[m] == B.ptr[m] && P.idx[m] == B.idx[m] && same_frs(P.val[m], B.val[m]);
            CHECK(same, "%s #%zu: parallel gate readers: R1CS rows / wires differ", what, id);
        }
    }
    if (rc != ZK_OK || rd != ZK_OK) return;
    bool same = B.n_public == D.n_public && same_frs(B.wires, D.wires);
    for (int m = 0; m < 3; m++) same = same && same_words(B.ptr[m], D.ptr[m]) && same_words(B.idx[m], D.idx[m]) && same_frs(B.val[m], D.val[m]);
    CHECK(same, "%s #%zu: R1CS rows / wires differ", what, id);
    // what a resident circuit keeps to assemble the wire vector on the device: the same wires from (values, order, products), and the circuit alone (no values
    // decoded) is the same circuit
    CHECK(B.wires.size() == 1 + B.n_values + B.prod_a.size() && B.order.size() == B.n_values, "%s #%zu: wire count", what, id);
    if (B.values_at) {
        CHECK(B.values_at + B.values_len <= text.size() && B.values_len == 8 + 64 * B.n_values && text[B.values_at - 1] == '"' && text[B.values_at + B.values_len] == '"',
              "%s #%zu: values span", what, id);
        std::vector<zkmi::HFr> w(B.wires.size());
        w[0] = zkmi::HFr::one();
        bool ok = true;
        for (size_t k = 0; k < B.n_values && ok; k++) {
            uint64_t t[4];
            ok = zkmi::felt_from_hex(text.data() + B.values_at + 8 + 64 * (size_t)B.order[k], 64, t);
            w[1 + k] = zkmi::HFr{{t[0], t[1], t[2], t[3]}}.to_mont();
        }
        for (size_t j = 0; j < B.prod_a.size() && ok; j++) {
            ok = B.prod_a[j] >= 1 && B.prod_a[j] <= B.n_values && B.prod_b[j] >= 1 && B.prod_b[j] <= B.n_values;
            if (ok) w[1 + B.n_values + j] = w[B.prod_a[j]] * w[B.prod_b[j]];
        }
        CHECK(ok && same_frs(w, B.wires), "%s #%zu: wires reassembled from order / products differ", what, id);
    }
    zkmi::RawR1CSBuilt C;
    std::string ec;
    const int rcc = zkmi::raw_r1cs_build(text.data(), text.size(), &C, &ec, false);
    bool same_c = rcc == ZK_OK && C.n_public == B.n_public && C.order == B.order && C.prod_a == B.prod_a && C.prod_b == B.prod_b && C.wires.empty();
    for (int m = 0; m < 3; m++) same_c = same_c && C.ptr[m] == B.ptr[m] && C.idx[m] == B.idx[m] && same_frs(C.val[m], B.val[m]);
    CHECK(same_c, "%s #%zu: circuit-only mode differs (rc %d %s)", what, id, rcc, ec.c_str());
}

// the shim's own text helpers (goffi.cpp) and the key / SRS header readers: no second implementation to compare with -- the sanitizers are the check,
// plus the invariants a caller relies on (an accepted header's sizes are consistent with the length it was given)
static void check_text(const std::string& text, size_t id) {
    std::vector<uint8_t> bytes;
    const bool ok = zkmi::hex_to_bytes(text.data(), text.size(), &bytes);
    CHECK(!ok || bytes.size() * 2 == text.size(), "hex_to_bytes #%zu: length", id);
    CHECK((text.size() & 1) || ok == zkmi::all_hex(text.data(), text.size()), "all_hex #%zu disagrees with hex_to_bytes", id);
    const char* u;
    size_t un;
    zkmi::unquote(text.data(), text.size(), &u, &un);
    CHECK(un <= text.size(), "unquote #%zu", id);
    std::vector<std::vector<uint8_t>> felts;
    if (zkmi::felts_from_hex(u, un, &felts)) {
        CHECK(un == 8 + 64 * felts.size(), "felts_from_hex #%zu: count", id);
        std::vector<zk_fr> m;
        (void)zkmi::be_to_mont(felts, &m);
    }
    size_t n = 0;
    if (zkmi::count_from_hex(text.data(), text.size(), &n)) CHECK(text.size() >= 8, "count_from_hex #%zu", id);
}
static void check_key_headers(const std::string& img, size_t id) {
    for (int is_hex = 0; is_hex < 2; is_hex++) {
        zkmi::PlonkKeyHeader h;
        std::string err;
        for (size_t nc : {(size_t)0, (size_t)3, (size_t)1000}) {
            const int rc = zkmi::plonk_pk_header(img.data(), img.size(), is_hex, nc + 5, nc, &h, &err);
            if (rc == ZK_OK) {
                const size_t nbytes = is_hex ? img.size() / 2 : img.size();
                CHECK(nbytes == zkmi::PLONK_PK_HEAD + 9 * (4 + 32 * h.n) + 24 * h.n && h.n == ((size_t)1 << h.logn), "plonk header #%zu accepted with inconsistent sizes", id);
            }
        }
        zkmi::SrsHeader sh;
        if (zkmi::kzg_srs_header(img.data(), img.size(), is_hex, &sh, &err) == ZK_OK) {
            const size_t nbytes = is_hex ? img.size() / 2 : img.size();
            CHECK(nbytes == 4 + 32 * sh.n_g1 + 128, "SRS header #%zu accepted with inconsistent sizes", id);
        }
        zkmi::Groth16KeyHeader gh;
        if (zkmi::groth16_pk_header(img.data(), img.size(), is_hex, &gh, &err) == ZK_OK) {
            const size_t nbytes = is_hex ? img.size() / 2 : img.size();
            CHECK(gh.total_bytes == nbytes, "Groth16 key header #%zu accepted with inconsistent sizes", id);
        }
    }
}

static bool ends_with(const std::string& s, const char* suf) { const size_t k = strlen(suf); return s.size() >= k && !s.compare(s.size() - k, k, suf); }

// The content key (acir_host.hpp: a keyed SipHash-2-4-128 tree) under the sanitizers: the function's published vectors, and the properties the caches rely on -- equal
// texts give equal keys; a one-byte change anywhere (first / last byte, either side of every leaf and segment boundary, the short last segment) gives another key;
// so does a text that differs only by trailing zero bytes, or by its length alone; a text cut at a boundary is not a prefix collision of the longer one.
static void content_key_checks() {
    if (!zkmi::siphash_selftest()) { fprintf(stderr, "SipHash self-test failed\n"); exit(1); }
    const size_t SEG = (size_t)1 << 16, LEAF = SEG / 8;
    for (size_t n : {(size_t)0, (size_t)1, (size_t)7, (size_t)8, LEAF - 1, LEAF, LEAF + 1, SEG - 1, SEG, SEG + 1, 3 * SEG + 5 * LEAF + 3, 70 * SEG + 17}) {  // (>= 64 segments: the threaded path)
        std::string t(n, '\0');
        for (size_t i = 0; i < n; i++) t[i] = (char)('a' + (i * 2654435761u >> 13) % 23);
        const zkmi::ContentKey k0 = zkmi::content_key(t.data(), n);
        if (!(k0 == zkmi::content_key(std::string(t).data(), n))) { fprintf(stderr, "content key: equal texts, different keys (n = %zu)\n", n); exit(1); }
        std::vector<size_t> at = {0, n ? n - 1 : 0};
        for (size_t b = LEAF; b < n; b += LEAF) { at.push_back(b - 1); at.push_back(b); }
        if (at.size() > 400) at.resize(400);
        for (size_t i : at) {
            if (i >= n) continue;
            std::string u = t;
            u[i] ^= 1;
            if (k0 == zkmi::content_key(u.data(), n)) { fprintf(stderr, "content key: byte %zu of %zu does not matter\n", i, n); exit(1); }
        }
        std::string z = t + std::string(1, '\0');
        if (k0 == zkmi::content_key(z.data(), n + 1)) { fprintf(stderr, "content key: a trailing zero byte does not matter (n = %zu)\n", n); exit(1); }
        if (n > 1) {
            zkmi::ContentKey kp = zkmi::content_key(t.data(), n - 1);
            if (kp.h[0] == k0.h[0] && kp.h[1] == k0.h[1]) { fprintf(stderr, "content key: a prefix shares the digest (n = %zu)\n", n); exit(1); }
        }
    }
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: parser_fuzz <cases> <seed file>...\n"); return 2; }
    content_key_checks();
    const size_t cases = (size_t)strtoull(argv[1], nullptr, 10);
    std::vector<std::pair<std::string, std::string>> seeds;
    for (int i = 2; i < argc; i++) {
        FILE* f = fopen(argv[i], "rb");
        if (!f) { fprintf(stderr, "cannot read %s\n", argv[i]); return 2; }
        std::string s;
        char buf[1 << 16];
        size_t k;
        while ((k = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, k);
        fclose(f);
        seeds.emplace_back(argv[i], s);
    }
    size_t done = 0, accepted = 0;
    const size_t nvals[] = {0, 1, 5, 6, 7, 300};
    for (size_t c = 0; c < cases; c++) {
        const auto& sd = seeds[c % seeds.size()];
        const std::string m = c < seeds.size() ? sd.second : mutate(sd.second);  // the seeds themselves first
        if (ends_with(sd.first, ".acir.json")) {
            const size_t nv = nvals[below(sizeof nvals / sizeof *nvals)];
            const int layout = (int)below(2);
            check_acir(m, nv, layout, "ACIR", c);
            zkmi::Gates G;
            std::string e;
            accepted += zkmi::lower_acir(m.data(), m.size(), nv, layout, false, &G, &e) == ZK_OK;
        } else if (ends_with(sd.first, ".raw.json")) {
            check_raw(m, "RawR1CS", c);
        } else if (ends_with(sd.first, ".bin")) {
            check_key_headers(m, c);
        } else {
            check_text(m, c);
        }
        done++;
    }
    printf("{\"cases\": %zu, \"accepted_acir\": %zu, \"mismatches\": %d}\n", done, accepted, failures);
    return failures ? 1 : 0;
}
